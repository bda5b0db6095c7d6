"""gpurun helper: K17's GELU epilogues on the Swin MLP shapes and K13's row sums (VALU-bound epilogues)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mask_bev_amd import ops
from _timeit import timeit
dev = torch.device('cuda', 0); dt = torch.bfloat16
T, C = 65536, 192
for s in range(4):
    x = torch.randn(T, C, device=dev).to(dt); w1 = (torch.randn(4 * C, C, device=dev) * 0.05).to(dt); b1 = torch.randn(4 * C, device=dev)
    g = torch.randn(T, C, device=dev).to(dt); w2 = (torch.randn(C, 4 * C, device=dev) * 0.05).to(dt); h = torch.randn(T, 4 * C, device=dev).to(dt)
    cs = torch.zeros(4 * C, device=dev)
    f = [timeit(lambda: ops.gemm16_nt(x, w1, b1, act=a, want_pre=a is not None)) for a in (None, 'gelu')]
    d = [timeit(lambda: ops.gemm16_nn(g, w2, act=a, aux=h if a else None, colsum=cs if a else None)) for a in (None, 'gelu')]
    print(f'stage {s + 1} T={T:6d} C={C:5d}  fc1 none/gelu = {f[0]:6.1f} {f[1]:6.1f} us   dgrad none/gelu\' = {d[0]:6.1f} {d[1]:6.1f} us', flush=True)
    T //= 4; C *= 2
x = (torch.randn(4000, 12544, device=dev) * 4).requires_grad_(); t = (torch.rand(4000, 12544, device=dev) > 0.7).float()
print('mask_loss_rows fwd %.1f us' % timeit(lambda: ops.mask_loss_rows(x.detach(), t)))
g4 = torch.randn(4000, 4, device=dev)
print('mask_loss_rows bwd %.1f us' % timeit(lambda: ops._lib.load() and ops.mask_loss_rows_backward(x.detach(), t, g4)) if hasattr(ops, 'mask_loss_rows_backward') else 'no direct bwd wrapper')
