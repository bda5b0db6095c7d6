"""gpurun helper: the (rows, n, dtype) entries of every grouped column-sum launch of one eager bench step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mask_bev_amd import synthetic, ops, _lib, tuning
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda:0')
kw = synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype='bf16')
tuning.use_tuned_gemms()
torch.manual_seed(0)
m = MaskBevModule(**kw).to(dev).train(); m.log_scalars = False
arena = m.flatten_parameters()
batch = synthetic.make_batch('semantic_kitti_512', 4, 0, 0, dev)
lib = _lib.load()
def hook(name, fn, args):
    if name == 'mbv_colsum_accum_group':
        cnt = args[6]
        ent = sorted(((int(args[2][i]), int(args[3][i]), int(args[1][i])) for i in range(cnt)), key=lambda e: -e[0] * e[1] * (2 if e[2] else 4))
        tot = sum(r * n * (2 if d else 4) for r, n, d in ent) / 1e6
        print(f'colsum group: {cnt} entries, {tot:.0f} MB; largest:', [(r, n, 'lo' if d else 'f32', round(r * n * (2 if d else 4) / 1e6, 1)) for r, n, d in ent[:12]], flush=True)
    return fn(*args)
for it in range(2):
    lib.hook = hook if it == 1 else None
    loss = m.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    arena.zero_grad()
