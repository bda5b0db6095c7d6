"""gpurun helper: K8 backward — the plain f32 form against the stack form (permuted rows; f32 / bf16 / fp16 stores)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mask_bev_amd import ops, _lib
from _timeit import timeit
dev = torch.device('cuda', 0)
lib = _lib.load()
d, b, q, h, w, p = 10, 4, 100, 128, 128, 12544
n = d * b * q
idx = torch.arange(n, dtype=torch.int32, device=dev)
coords = torch.rand(d * b, p, 2, device=dev)
cidx = (idx // q).to(torch.int32)
gout = torch.randn(n, p, device=dev)
gs = torch.empty(n, h, w, device=dev)
P, S = ops._ptr, ops._stream
print('plain f32   %.1f us' % timeit(lambda: lib.mbv_point_sample_bwd(P(gout), P(idx), P(coords), P(cidx), n, p, h, w, n, P(gs), S())))
for dt in (torch.float32, torch.bfloat16, torch.float16):
    out = torch.empty(b, d, q, h * w, dtype=dt, device=dev)
    print('stack', dt, '%.1f us' % timeit(lambda: lib.mbv_point_sample_bwd_stack(P(gout), P(idx), P(coords), P(cidx), n, p, h, w, d, b, q, P(out), ops._dt_flag(dt), S())))
