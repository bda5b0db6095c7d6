import torch, numpy as np
from scipy.optimize import linear_sum_assignment
from mask_bev_amd import ops
dev='cuda'
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
g=torch.Generator().manual_seed(0)
rand=torch.rand(40,100,100,generator=g)*10
base=torch.rand(40,100,30,generator=g)*10
padded=torch.cat([base, base[:,:,:1].expand(-1,-1,70)+0.0],2).contiguous()
for name,c in [('random 100x100',rand),('30 real + 70 identical cols',padded)]:
    cd=c.to(dev)
    print(name,'%.0f us'%t(lambda: ops.hungarian(cd)))
    got=ops.hungarian(cd).cpu().numpy()
    for i in range(3):
        r,cc=linear_sum_assignment(c[i].numpy())
        print('   cost ours %.6f scipy %.6f'%(c[i].numpy()[np.arange(100),got[i]].sum(), c[i].numpy()[r,cc].sum()))
