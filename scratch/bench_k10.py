import torch
from mask_bev_amd import ops
dev='cuda'
def t(fn,n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
R,n,k,H,W,nr=4000,37632,9408,128,128,3136
src=torch.randn(R,H,W,device=dev)*0.05
idx=torch.arange(R,device=dev,dtype=torch.int32)
coords=torch.rand(R,n,2,device=dev); rc=torch.rand(R,nr,2,device=dev)
print("fused %.0f us"%t(lambda: ops.sample_select_uncertain(src,idx,coords,k,rc)))
seed=torch.tensor([12345],dtype=torch.int64,device=dev)
print("fused rng %.0f us"%t(lambda: ops.sample_select_uncertain(src,idx,None,k,rc,seed=seed,num_candidates=n)))
rows=torch.arange(R,device=dev,dtype=torch.int32)
lg=ops.point_sample(src,idx,coords,rows)
print('K8 %.0f us'%t(lambda: ops.point_sample(src,idx,coords,rows)), 'K10 %.0f us'%t(lambda: ops.select_uncertain_points(lg,coords,k)))
