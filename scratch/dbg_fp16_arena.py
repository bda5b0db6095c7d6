"""Which parameters differ most between the arena path and plain autograd, per compute dtype and loss scale."""
import sys
import torch
sys.path.insert(0, '.')
from mask_bev_amd.mask_bev_module import MaskBevModule
from tests.util_cfg import random_gt, random_scans, tiny_kwargs

dev = torch.device('cuda', 0)
for dtype, ls in [('bf16', 1.0), ('fp16', 1.0), ('fp16', 64.0), ('fp16', 1024.0), ('fp16', 8192.0)]:
    kw = tiny_kwargs(nx=96, ny=96, q=8)
    kw['compute_dtype'] = dtype
    torch.manual_seed(0)
    plain = MaskBevModule(**kw).to(dev).train()
    torch.manual_seed(0)
    flat = MaskBevModule(**kw).to(dev).train()
    flat.load_state_dict(plain.state_dict())
    flat.flatten_parameters()
    for m in (plain, flat):
        m.log_scalars = False
        h = m._panoptic_head._panoptic_head
        h.num_points = 600
        h.point_seed = 5
    scans = [x.to(dev) for x in random_scans(kw, [3000, 2500], seed=0)]
    labels, gt = random_gt(kw, 2, 3, seed=10)
    batch = (scans, (labels.to(dev), gt.to(dev)))
    l0 = plain.training_step(batch, 0)
    (l0 * ls).backward()
    l1 = flat.training_step(batch, 0)
    (l1 * ls).backward()
    worst = []
    for (n, p), (_, q) in zip(plain.named_parameters(), flat.named_parameters()):
        if p.grad is None:
            continue
        scale = float(p.grad.abs().max()) + 1e-9
        err = float((p.grad.float() - q.grad).abs().max()) / scale
        l2 = float((p.grad.float() - q.grad).norm() / (q.grad.norm() + 1e-20))
        fin = bool(torch.isfinite(p.grad).all()) and bool(torch.isfinite(q.grad).all())
        worst.append((err, l2, fin, n[-70:]))
    worst.sort(reverse=True, key=lambda t: (t[0] if t[0] == t[0] else 1e9))
    print(dtype, ls, float(l0), float(l1))
    for w in worst[:6]:
        print('   ', w)
