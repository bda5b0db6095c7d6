"""Is the forward bit-reproducible per compute dtype?  Two forwards of one module on one batch, stage by stage."""
import sys
import torch
sys.path.insert(0, '.')
from mask_bev_amd.mask_bev_module import MaskBevModule
from tests.util_cfg import random_scans, tiny_kwargs

dev = torch.device('cuda', 0)
for dtype in ('bf16', 'fp16'):
    kw = tiny_kwargs(nx=96, ny=96, q=8)
    kw['compute_dtype'] = dtype
    torch.manual_seed(0)
    m = MaskBevModule(**kw).to(dev).eval()
    scans = [x.to(dev) for x in random_scans(kw, [3000, 2500], seed=0)]
    outs = []
    for rep in range(3):
        with torch.no_grad(), m._autocast():
            x = m._encoder(scans)
            feats = m._backbone(x)
            cls, masks, _ = m._panoptic_head(feats)
        outs.append([x] + list(feats) + list(cls) + list(masks))
    names = ['enc'] + [f'feat{i}' for i in range(4)] + [f'cls{i}' for i in range(10)] + [f'mask{i}' for i in range(10)]
    bad = [(n, float((a.float() - b.float()).abs().max())) for n, a, b in zip(names, outs[0], outs[1]) if not torch.equal(a, b)]
    bad2 = [(n, float((a.float() - b.float()).abs().max())) for n, a, b in zip(names, outs[1], outs[2]) if not torch.equal(a, b)]
    print(dtype, 'rep0 vs rep1:', bad[:6], '| rep1 vs rep2:', bad2[:6])
