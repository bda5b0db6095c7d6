import torch, types
from mask_bev_amd import ops
dev='cuda'
B,C,ny,nx=4,128,512,512
cells=ny*nx
torch.manual_seed(0)
per=[20000,21000,19000,20500]
c2p=torch.full((B,cells),-1,dtype=torch.int32,device=dev); starts=[0]
off=0
for b,v in enumerate(per):
    idx=torch.randperm(cells,device=dev)[:v].sort().values
    c2p[b,idx]=torch.arange(off,off+v,dtype=torch.int32,device=dev); off+=v; starts.append(off)
V=off
p=types.SimpleNamespace(cell_to_pillar=c2p,pillar_batch_start=torch.tensor(starts,dtype=torch.int32,device=dev))
feats=torch.randn(V,C,device=dev,requires_grad=True)
w=torch.randn(C,ny,nx,device=dev,requires_grad=True); bias=torch.randn(C,ny,nx,device=dev,requires_grad=True)
g=torch.randn(B,C,ny,nx,device=dev)
ops.TIMER.reset(); ops.TIMER.enabled=True
for _ in range(12):
    y=ops.scatter_layernorm(feats,w,bias,p,B,ny,nx,1e-3)
    y.backward(g)
    feats.grad=None; w.grad=None; bias.grad=None
torch.cuda.synchronize()
for k,v in ops.TIMER.summary_ms().items():
    v=v[2:]; print(k,'avg %.1f us'%(sum(v)/len(v)*1e3))
