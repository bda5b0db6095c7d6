"""Cost of K17's GELU / GELU' epilogues on the Swin MLP shapes (graph-timed): act None / relu / gelu, forward and dgrad."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mask_bev_amd import ops, tuning
from _timeit import timeit
dev = torch.device('cuda', 0); dt = torch.bfloat16
tuning.use_tuned_gemms()
T, C = 65536, 192
for s in range(4):
    x = torch.randn(T, C, device=dev).to(dt); w1 = (torch.randn(4 * C, C, device=dev) * 0.05).to(dt); b1 = torch.randn(4 * C, device=dev)
    g = torch.randn(T, C, device=dev).to(dt); w2 = (torch.randn(C, 4 * C, device=dev) * 0.05).to(dt); h = torch.randn(T, 4 * C, device=dev).to(dt)
    cs = torch.zeros(4 * C, device=dev)
    f = [timeit(lambda: ops.gemm16_nt(x, w1, b1, act=a, want_pre=a is not None)) for a in (None, 'relu', 'gelu')]
    d = [timeit(lambda: ops.gemm16_nn(g, w2, act=a, aux=h if a else None, colsum=cs if a else None)) for a in (None, 'relu', 'gelu')]
    print(f'stage {s + 1} T={T:6d} C={C:5d}  fc1 none/relu/gelu = {f[0]:6.1f} {f[1]:6.1f} {f[2]:6.1f} us   dgrad none/relu\'/gelu\' = {d[0]:6.1f} {d[1]:6.1f} {d[2]:6.1f} us', flush=True)
    T //= 4; C *= 2
