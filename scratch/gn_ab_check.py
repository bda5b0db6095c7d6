"""K18 on / off inside the whole model (tiny configuration, 16-bit compute): difference of the pixel decoder's outputs
and of the per-layer logits between the two paths — rounding-level, or a discontinuity downstream?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from util_cfg import tiny_kwargs, random_scans  # noqa: E402
from mask_bev_amd.mask_bev_module import MaskBevModule  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import maskbev_oracle as O  # noqa: E402  (checker only: state dict generator)

dev = torch.device('cuda', 0)
for dtype in ('fp16', 'bf16'):
    kw = dict(tiny_kwargs(), compute_dtype=dtype)
    cfg = O.make_cfg(**tiny_kwargs())
    sd = O.make_state_dict(cfg, 7)
    m = MaskBevModule(**kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(dev).train()
    scans = [s.to(dev) for s in random_scans(tiny_kwargs(), [3000, 2000], seed=2)]
    pd = m._panoptic_head._panoptic_head.pixel_decoder
    grabbed = {}
    h = pd.register_forward_hook(lambda mod, inp, out: grabbed.__setitem__('out', out))
    res = {}
    for mode in ('0', '1'):
        os.environ['MBV_GROUPNORM'] = mode
        with torch.no_grad():
            cls, masks, _ = m(scans)
        mf, outs = grabbed['out']
        res[mode] = ([mf.float().clone()] + [o.float().clone() for o in outs], [x.float().clone() for x in masks],
                     [x.float().clone() for x in cls])
    h.remove()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))
    print(dtype, 'pixel decoder outputs (mask_feature, memory levels):', [f'{rel(a, b):.2e}' for a, b in zip(res['1'][0], res['0'][0])])
    print(dtype, 'mask logits per decoder output:', [f'{rel(a, b):.1e}' for a, b in zip(res['1'][1], res['0'][1])])
    print(dtype, 'class logits per decoder output:', [f'{rel(a, b):.1e}' for a, b in zip(res['1'][2], res['0'][2])])
