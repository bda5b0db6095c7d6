import torch
from mask_bev_amd import ops
dev='cuda'
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for T,N in [(21504,256),(21504,1024),(65536,128),(65536,384),(65536,512),(16384,256),(16384,1024),(4096,2048),(400,256),(400,2048),(1024,1536)]:
    g=torch.randn(T,N,device=dev).bfloat16(); out=torch.zeros(N,device=dev)
    a=t(lambda: ops.colsum_accum(g,out)); b=t(lambda: g.sum(0,dtype=torch.float32))
    print(T,N,'mine %.1f us torch %.1f us'%(a,b), 'GB/s mine %.0f'%(T*N*2/a/1e3))
