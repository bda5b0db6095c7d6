import time, numpy as np, torch
from mask_bev_amd import batch
from oracle import batch_oracle as BO
dev='cuda'
rng=np.random.default_rng(0)
maps=np.zeros((4,512,512),dtype=np.int64)
for b in range(4):
    for k in range(35):
        x0,y0=rng.integers(0,480,2); maps[b,x0:x0+rng.integers(4,30),y0:y0+rng.integers(4,30)]=1000+k
t=torch.from_numpy(maps).to(dev)
def tm(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n*1e3
print('K14 f32 masks   %.0f us / batch of 4 (writes 420 MB)'%tm(lambda: batch.instance_targets(t,100,10)))
print('K14 packed masks %.0f us / batch of 4 (writes 13 MB)'%tm(lambda: batch.instance_targets(t,100,10,packed=True)))
h2d_dense=torch.zeros(4,100,512,512).pin_memory(); h2d_map=torch.from_numpy(maps.astype(np.int32)).pin_memory()
print('H2D dense masks %.0f us, H2D instance maps %.0f us'%(tm(lambda: h2d_dense.to(dev,non_blocking=True),5), tm(lambda: h2d_map.to(dev,non_blocking=True),5)))
t0=time.perf_counter(); 
for b in range(4): BO.instance_targets(maps[b],100,10)
print('oracle (numpy, 1 core) %.0f ms / batch of 4'%((time.perf_counter()-t0)*1e3))
