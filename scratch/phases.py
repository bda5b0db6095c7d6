import sys, torch, time
sys.path.insert(0, '.')
from mask_bev_amd import synthetic
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda:0')
kw = synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype='bf16')
m = MaskBevModule(**kw).to(dev).train(); m.log_scalars = False
opt = m.configure_optimizers()['optimizer']
pool = [synthetic.make_batch('semantic_kitti_512', 4, 0, s, dev) for s in range(2)]
def ev(): e = torch.cuda.Event(enable_timing=True); e.record(); return e
acc = {}
for it in range(8):
    scans, (labels, masks) = pool[it % 2]
    t = [ev()]
    with m._autocast():
        x = m._encoder(scans); t.append(ev())
        f = m._backbone(x); t.append(ev())
        cls, mk, _ = m._panoptic_head(f); t.append(ev())
    ld = m.compute_loss(cls, mk, labels, masks); loss = m.loss(ld); t.append(ev())
    loss.backward(); t.append(ev())
    opt.step(); opt.zero_grad(set_to_none=True); t.append(ev())
    torch.cuda.synchronize()
    if it >= 3:
        for name, a, b in zip(['encoder', 'backbone', 'head', 'loss', 'backward', 'optimizer'], t[:-1], t[1:]):
            acc[name] = acc.get(name, 0) + a.elapsed_time(b) / 5
print({k: round(v, 2) for k, v in acc.items()}, 'total', round(sum(acc.values()), 2))
