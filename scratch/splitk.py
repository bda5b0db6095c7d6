import torch, time
dev = torch.device('cuda:0')
def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for (T, cin, cout) in [(65536, 192, 576), (65536, 192, 768), (65536, 768, 192), (65536, 192, 192), (16384, 384, 1152), (16384, 384, 1536),
                       (4096, 768, 2304), (4096, 768, 3072), (1024, 1536, 6144), (21504, 256, 1024), (21504, 256, 256), (65536, 2048, 192)]:
    x = torch.randn(T, cin, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(T, cout, device=dev, dtype=torch.bfloat16)
    t0 = bench(lambda: dy.t().mm(x))
    res = [f'mm {t0:7.1f}us']
    for S in (8, 32, 128):
        if T % S: continue
        xs, ds = x.view(S, T // S, cin), dy.view(S, T // S, cout)
        t1 = bench(lambda: torch.bmm(ds.transpose(1, 2), xs).sum(0))
        res.append(f'S={S}: {t1:7.1f}us')
    ts = bench(lambda: dy.sum(0))
    print(T, cin, cout, ' | '.join(res), f'| bias-sum {ts:6.1f}us', f'| ideal@500TF {2*T*cin*cout/500e12*1e6:5.1f}us')
