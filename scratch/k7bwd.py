import torch
dev = torch.device('cuda:0')
def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
B, Q, C, HW = 4, 100, 256, 16384
dl = torch.randn(B, Q, HW, device=dev, dtype=torch.bfloat16)
ff = torch.randn(B, C, HW, device=dev, dtype=torch.bfloat16)
e = torch.randn(B, Q, C, device=dev, dtype=torch.bfloat16)
print('dE matmul', bench(lambda: torch.matmul(dl, ff.transpose(1, 2))))
print('dF matmul', bench(lambda: torch.matmul(e.transpose(1, 2), dl)))
for S in (8, 16, 32):
    hw = HW // S
    f = lambda: torch.matmul(dl.view(B, Q, S, hw).transpose(1, 2), ff.view(B, C, S, hw).permute(0, 2, 3, 1)).sum(1, dtype=torch.float32)
    print('dE split', S, bench(f))
    f2 = lambda: torch.einsum('bqsp,bcsp->bsqc', dl.view(B, Q, S, hw), ff.view(B, C, S, hw)).sum(1, dtype=torch.float32)
    print('dE einsum split', S, bench(f2))
r1 = torch.matmul(dl, ff.transpose(1, 2)).float()
r2 = torch.matmul(dl.view(B, Q, 16, HW // 16).transpose(1, 2), ff.view(B, C, 16, HW // 16).permute(0, 2, 3, 1)).sum(1, dtype=torch.float32)
print('max rel diff', ((r1 - r2).abs().max() / r1.abs().max()).item())
