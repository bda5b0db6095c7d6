import sys, torch
sys.path.insert(0, '.')
from tests.util_cfg import random_gt, random_scans, tiny_kwargs
from mask_bev_amd.graph import GraphedTrainStep
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda:0')
mode = sys.argv[1]
torch.manual_seed(0)
kw = tiny_kwargs(nx=96, ny=96, q=8)
if 'bf16' in mode: kw['compute_dtype'] = 'bf16'
m = MaskBevModule(**kw).to(dev).train(); m.log_scalars = False
m._panoptic_head._panoptic_head.num_points = 2000
scans = [x.to(dev) for x in random_scans(kw, [3000, 2500], seed=0)]
labels, gt = random_gt(kw, 2, 3, seed=10)
batch = (scans, (labels.to(dev), gt.to(dev)))
if 'eagerfwd' in mode:
    with torch.no_grad(): l = m.training_step(batch, 0)
    torch.cuda.synchronize(); print('eager fwd ok', flush=True)
elif 'eagerhead' in mode:
    x = m._encoder(scans).detach().requires_grad_()
    cls, mk, _ = m._panoptic_head(m._backbone(x)); l = m.loss(m.compute_loss(cls, mk, batch[1][0], batch[1][1])); l.backward(); m.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); print('eager head ok', flush=True)
elif 'eagerenc' in mode:
    x = m._encoder(scans); x.sum().backward(); m.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); print('eager enc ok', flush=True)
elif 'eager' in mode:
    l = m.training_step(batch, 0); l.backward(); m.zero_grad(set_to_none=True); torch.cuda.synchronize(); print('eager ok', flush=True)
class NoOpt:
    def step(self): pass
g = GraphedTrainStep(m, NoOpt(), batch)
print('capture ok', flush=True)
print(float(g.step(batch)))
