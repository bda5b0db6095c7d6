import sys, os, torch
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
seen = {}
for it in range(12):
    junk = [torch.full((n,), float('nan') if it % 2 else 1e30, device=dev) for n in (1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16)]
    del junk
    m, cfg, sd = _build(kw, dev, seed=7)
    head = m._panoptic_head._panoptic_head; head.num_points = 256; head.point_seed = 11
    pd = head.pixel_decoder
    snaps = {}
    def mk(name):
        def hook(mod, inp, out):
            torch.cuda.synchronize()
            o = out[0] if isinstance(out, tuple) else out
            snaps[name] = o.detach().float().clone()
            if name == 'pd':
                snaps['pd_tail'] = [t.detach().float().clone() for t in out[1]]
        return hook
    hs = [pd.lateral_convs[0].register_forward_hook(mk('lateral')), pd.output_convs[0].register_forward_hook(mk('outconv')),
          pd.register_forward_hook(mk('pd')), pd.encoder.layers[-1].register_forward_hook(mk('enc_last'))]
    m.train()
    loss = m.training_step(([s.to(dev) for s in scans], (labels.to(dev), gt.to(dev))), 1)
    torch.cuda.synchronize()
    key = '%.8f' % float(loss.detach())
    if key not in seen:
        seen[key] = snaps
        print(key)
if len(seen) == 2:
    a, b = list(seen.values())
    for k in a:
        if isinstance(a[k], list):
            for j, (u, v) in enumerate(zip(a[k], b[k])): print(k, j, float((u - v).abs().max()))
        else:
            d = (a[k] - b[k]).abs(); print(k, tuple(a[k].shape), float(d.max()), 'per-image', [float(d[i].max()) for i in range(d.shape[0])])
print('outcomes', len(seen))
