import sys, torch, time
sys.path.insert(0, '.')
from mask_bev_amd import synthetic
from mask_bev_amd.mask_bev_module import MaskBevModule
dev = torch.device('cuda:0')
kw = synthetic.module_kwargs('semantic_kitti_512', 4, compute_dtype='bf16')
m = MaskBevModule(**kw).to(dev).train(); m.log_scalars = False
opt = m.configure_optimizers()['optimizer']
pool = [synthetic.make_batch('semantic_kitti_512', 4, 0, s, dev) for s in range(2)]
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = m.training_step(pool[i % 2], i); loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('steps ms:', ' '.join(f'{t:.1f}' for t in ts))
s = sorted(ts[5:]); print('min', s[0], 'median', s[len(s)//2], 'max', s[-1])
