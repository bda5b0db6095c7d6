"""Does hipBLASLt fuse bias + ReLU into the few-row f32 GEMMs of the decoder (torch._addmm_activation)?  Kernel count and
graph-timed duration of linear + relu against the fused call, 400 x 256 -> 2048 (the decoder FFN) and 400 x 256 -> 256."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm_util import timeit
dev = torch.device('cuda', 0)
for n in (2048, 256):
    x = torch.randn(400, 256, device=dev)
    w = torch.randn(n, 256, device=dev) * 0.05
    b = torch.randn(n, device=dev)
    f1 = lambda: torch.relu(torch.nn.functional.linear(x, w, b))
    f2 = lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)
    assert torch.allclose(f1(), f2(), rtol=1e-5, atol=1e-5)
    print(n, 'linear + relu us:', round(timeit(f1), 2), ' _addmm_activation us:', round(timeit(f2), 2))
