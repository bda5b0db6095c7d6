import sys, ctypes, os, torch
sys.path.insert(0, '.')
from mask_bev_amd import synthetic, decoder_fused as DF, _lib
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = MaskBevModule(**synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype='bf16')).to(dev).train()
m.log_scalars = False; m.flatten_parameters()
batch = synthetic.make_batch('semantic_kitti_512', 4, 0, 0, dev)
raw = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), 'libmaskbev_hip.so'))
names = ['LOAD', 'STORE', 'GEMM', 'LN', 'LN_BWD', 'ADD', 'COLSUM', 'FFN', 'FFN_IO']
orig = DF.Program.run
seen = {}
def run(self):
    orig(self)
    seen[self.label] = seen.get(self.label, 0) + 1
    if seen[self.label] == 3:
        torch.cuda.synchronize()
        b = (ctypes.c_ulonglong * 160)(); raw.mbv_rowchain_debug(b); t = list(b)
        n = len(self.stages)
        out = []; prev = t[1]; i = 0
        while i < n:
            op = int(t[80 + i]); out.append('%s %.1f' % (names[op], (t[2 + i] - prev) / 100.0)); prev = t[2 + i]
            i += 2 if op == 7 else 1
        print(self.label, 'total %.1f us:' % ((prev - t[0]) / 100.0), ' | '.join(out))
DF.Program.run = run
loss = m.training_step(batch, 0); loss.backward(); torch.cuda.synchronize()
