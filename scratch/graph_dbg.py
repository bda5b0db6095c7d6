import sys, torch
sys.path.insert(0, '.')
from mask_bev_amd import synthetic
from mask_bev_amd.mask_bev_module import MaskBevModule
from mask_bev_amd.graph import GraphedTrainStep
dev = torch.device('cuda:0')
torch.manual_seed(0)
kw = synthetic.module_kwargs('semantic_kitti_512', 2, compute_dtype='bf16')
m = MaskBevModule(**kw).to(dev).train(); m.log_scalars = False
opt = m.configure_optimizers()['optimizer']
pool = [synthetic.make_batch('semantic_kitti_512', 2, 0, s, dev) for s in range(2)]
MODE = sys.argv[1] if len(sys.argv) > 1 else 'alt'
g = GraphedTrainStep(m, opt, pool[0])
if MODE == 'noopt':
    opt.step = lambda *a, **k: None
import types
orig_step = opt.step
for i in range(3):
    def dbg_step(*a, **k):
        bad = [(n, float(p.grad.float().abs().max())) for n, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        big = sorted([(float(p.grad.float().abs().max()), n) for n, p in m.named_parameters() if p.grad is not None])[-4:]
        print('  nonfinite grads:', bad[:6], ' largest:', big, flush=True)
        return orig_step(*a, **k)
    if MODE != 'noopt': opt.step = dbg_step
    loss = g.step(pool[0 if MODE == 'same' else i % 2]); torch.cuda.synchronize()
    gn = torch.stack([p.grad.float().norm() for p in m._backbone.parameters() if p.grad is not None]).norm()
    xg = g.x_static.grad
    print(i, 'loss', float(loss), 'backbone grad norm', float(gn), 'xgrad finite', bool(torch.isfinite(xg).all()), 'params finite',
          all(bool(torch.isfinite(p).all()) for p in m.parameters()), flush=True)
