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
orig = DF.Program.run
seen = set()
def run(self):
    orig(self)
    if self.label in ('B.fwd', 'B.bwd') and self.label not in seen:
        seen.add(self.label)
        torch.cuda.synchronize()
        b = (ctypes.c_ulonglong * 16)(); raw.mbv_rowchain_debug3(b); u = list(b)
        print(self.label, 'FFN op (wave 0): zero %.2f | phase 1 %.2f | phase 2 %.2f | wait others %.2f us' % tuple(x / 100.0 for x in (u[1]-u[0], u[2]-u[1], u[3]-u[2], u[4]-u[3])))
DF.Program.run = run
loss = m.training_step(batch, 0); loss.backward(); torch.cuda.synchronize()
