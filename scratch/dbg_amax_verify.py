"""Debug: consumed absmax records against fresh maxima, eager pass vs captured pass (switches.amax_verify)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mask_bev_amd import ops, switches, synthetic
from mask_bev_amd.graph import GraphedTrainStep
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda', 0)
torch.manual_seed(0)
wl, b = 'semantic_kitti_512', 2
m = MaskBevModule(**synthetic.module_kwargs(wl, b, compute_dtype='fp32')).to(dev).train()
m.log_scalars = False
m.flatten_parameters()
opt = m.configure_optimizers()['optimizer']
data = [synthetic.make_batch(wl, b, 0, s, dev) for s in range(3)]
switches.set_value('amax_verify', True)
for k in sys.argv[1:]:
    n, _, v = k.partition('=')
    switches.set_value(n, v)
g = GraphedTrainStep(m, opt, data[0], warmup_iters=1)
n_eager = sum(1 for e in ops.AMAX_VERIFY.entries if not e[2])
steps = int(os.environ.get('STEPS', '3'))
for s in range(steps):
    g.step(data[s % 3])
rep = ops.AMAX_VERIFY.report()
print('entries', len(rep), 'eager', n_eager)
bad = 0
for i, (w, s, c, t, r) in enumerate(rep):
    flag = '' if (r >= t and (t == 0 or r / t < 300)) else '   <<<<'
    bad += bool(flag)
    if flag or os.environ.get('ALL'):
        print(i, 'cap' if c else 'eag', w, s, f'true {t:.6g} rec {r:.6g}', flag)
print('bad', bad)
