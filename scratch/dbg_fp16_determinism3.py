import sys
import torch
sys.path.insert(0, '.')
from mask_bev_amd.mask_bev_module import MaskBevModule
from tests.util_cfg import random_scans, tiny_kwargs
dev = torch.device('cuda', 0)
kw = tiny_kwargs(nx=96, ny=96, q=8)
kw['compute_dtype'] = 'fp16'
torch.manual_seed(0)
m = MaskBevModule(**kw).to(dev).eval()
scans = [x.to(dev) for x in random_scans(kw, [3000, 2500], seed=0)]
rec = []
def hook(name):
    def f(mod, inp, out):
        def flat(o):
            if isinstance(o, torch.Tensor):
                return [o.detach().clone()]
            if isinstance(o, (tuple, list)):
                return [t for x in o for t in flat(x)]
            return []
        rec[-1].append((name, flat(inp), flat(out)))
    return f
for n, mod in m._panoptic_head.named_modules():
    mod.register_forward_hook(hook(n))
for rep in range(2):
    rec.append([])
    with torch.no_grad():
        m(scans)
a, b = rec
print(len(a), len(b))
shown = 0
for (n1, i1, o1), (n2, i2, o2) in zip(a, b):
    ieq = all(torch.equal(x, y) for x, y in zip(i1, i2))
    oeq = all(torch.equal(x, y) for x, y in zip(o1, o2))
    if not oeq:
        print(n1, 'inputs equal:', ieq, 'outputs equal:', oeq, [tuple(x.shape) for x in o1][:3], [x.dtype for x in o1][:3])
        shown += 1
        if shown > 6:
            break
