"""gpurun helper: is the forward loss of the tiny model bit-reproducible from run to run?  usage: python scratch/dbg_determinism.py [switch=value ...]"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from mask_bev_amd import switches
for kv in sys.argv[1:]:
    k, v = kv.split('='); switches.set_value(k, v)
from util_cfg import tiny_kwargs, random_scans, random_gt
from test_model_gpu import _build
dev = torch.device('cuda:0')
kw = tiny_kwargs(); kw['compute_dtype'] = 'bf16'
switches.set_value('decoder_fused', '0')
scans = random_scans(kw, [3000, 2000], seed=2); labels, gt = random_gt(kw, 2, 3, seed=4)
vals = []
import os
for it in range(8):
    if os.environ.get('POISON'):
        junk = [torch.full((n,), float('nan') if it % 2 else 1e30, device=dev) for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16)] 
        del junk
    m, cfg, sd = _build(kw, dev, seed=7)
    head = m._panoptic_head._panoptic_head; head.num_points = 256; head.point_seed = 11
    m.train()
    loss = m.training_step(([s.to(dev) for s in scans], (labels.to(dev), gt.to(dev))), 1)
    vals.append(float(loss.detach()))
print(sys.argv[1:], ['%.8f' % v for v in vals], 'distinct:', len(set(vals)))
