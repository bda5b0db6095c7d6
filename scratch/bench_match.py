"""gpurun helper: K13c (mbv_match_products) against the term planes + f32 GEMM it replaces, at the bench shape."""
import sys, time, torch
sys.path.insert(0, '.')
from mask_bev_amd import ops
dev = torch.device('cuda:0')
n, q, g, p = 40, 100, 100, 12544
x = torch.randn(n, q, p, device=dev) * 4
t = (torch.rand(n, g, p, device=dev) > 0.7).float()
def timeit(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
tt = t.transpose(1, 2)
def old():
    terms, sums = ops.match_cost_terms(x, ones_row=True)
    return torch.matmul(terms, tt)
print('terms + f32 GEMM: %.1f us' % timeit(old))
for s in (None, 6, 8, 12, 14, 19, 25):
    print('match_products splits', s, ': %.1f us' % timeit(lambda: ops.match_products(x, t, splits=s)))
