import torch, time
dev='cuda'
a=torch.randn(400,256,device=dev,dtype=torch.bfloat16); b=torch.randn(400,2048,device=dev,dtype=torch.bfloat16)
g=torch.zeros(2048,256,device=dev)
for name,fn in [('mm out_dtype', lambda: torch.mm(b.t(), a, out_dtype=torch.float32)),
                ('addmm out_dtype', lambda: torch.addmm(g, b.t(), a, out_dtype=torch.float32)),
                ('addmm out_dtype out=', lambda: torch.addmm(g, b.t(), a, out_dtype=torch.float32, out=g)),
                ('bmm out_dtype', lambda: torch.bmm(b.t()[None], a[None], out_dtype=torch.float32)),
                ('baddbmm out_dtype', lambda: torch.baddbmm(g[None], b.t()[None], a[None], out_dtype=torch.float32))]:
    try:
        r=fn(); torch.cuda.synchronize()
        ref=b.float().t()@a.float()
        print(name,'OK',r.dtype,(r.reshape(ref.shape)-ref).abs().max().item()/ref.abs().max().item())
    except Exception as e:
        print(name,'FAIL',type(e).__name__,str(e)[:150])
# fused adamw timing on a param set like the model's
ps=[torch.randn(33_554_432,device=dev,requires_grad=True) for _ in range(2)]+[torch.randn(200_000,device=dev,requires_grad=True) for _ in range(700)]
for p in ps: p.grad=torch.randn_like(p)
for kw in [dict(foreach=True),dict(fused=True)]:
    opt=torch.optim.AdamW(ps,lr=1e-4,weight_decay=0.01,**kw)
    for _ in range(3): opt.step()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): opt.step()
    e1.record(); torch.cuda.synchronize(); print(kw, e0.elapsed_time(e1)/5,'ms  params',sum(p.numel() for p in ps)/1e6,'M')
