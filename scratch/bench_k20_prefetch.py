"""Does the absmax pass in front of a K20 GEMM pay for itself as a prefetch?  x written by a LayerNorm launch, then
(a) absmax(x, w) + GEMM (the step's default), (b) GEMM with a precomputed record (no pass over x), graph-timed, LN time subtracted."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scratch._timeit import timeit
from mask_bev_amd import ops
dev = torch.device('cuda', 0)
for (m, k, n) in ((65536, 192, 576), (65536, 192, 768), (16384, 384, 1536), (4096, 768, 3072), (65536, 768, 192)):
    a32 = torch.randn(m, k, device=dev)
    g, b = torch.ones(k, device=dev), torch.zeros(k, device=dev)
    w = torch.randn(n, k, device=dev) * 0.05
    bias = torch.randn(n, device=dev)
    y0 = ops.add_layernorm(a32, None, g, b, 1e-5, torch.float32)
    rec = ops.f32_absmax([y0, w])
    ln = timeit(lambda: ops.add_layernorm(a32, None, g, b, 1e-5, torch.float32))
    def with_pass():
        y = ops.add_layernorm(a32, None, g, b, 1e-5, torch.float32)
        return ops.gemm32s_nt(y, w, bias)                       # absmax([y, w]) inside
    def without():
        y = ops.add_layernorm(a32, None, g, b, 1e-5, torch.float32)
        return ops.gemm32s_nt(y, w, bias, amax=rec)
    def warm():
        return ops.gemm32s_nt(y0, w, bias, amax=rec)
    def hinted():
        y = ops.add_layernorm(a32, None, g, b, 1e-5, torch.float32)
        return ops.gemm32s_nt(y, w, bias, amax=(rec[0:1], rec[1:2]), hint_out=True)
    def chain():
        y = ops.add_layernorm(a32, None, g, b, 1e-5, torch.float32)
        h = ops.gemm32s_nt(y, w, bias, amax=(rec[0:1], rec[1:2]), hint_out=True)
        return ops.gemm32s_nn(h, w, ops.amax_hint_get(h), rec[1:2], hint_out=True)
    def chain0():
        y = ops.add_layernorm(a32, None, g, b, 1e-5, torch.float32)
        h = ops.gemm32s_nt(y, w, bias, amax=(rec[0:1], rec[1:2]))
        return ops.gemm32s_nn(h, w, rec[0:1], rec[1:2])
    t1, t2, t3, t4, t5, t6 = timeit(with_pass), timeit(without), timeit(warm), timeit(hinted), timeit(chain), timeit(chain0)
    print(f'{m}x{k}->{n}: LN {ln:5.1f} | absmax + GEMM {t1 - ln:6.1f} | GEMM alone behind LN {t2 - ln:6.1f} | GEMM warm {t3:6.1f} | with amax_out {t4 - ln:6.1f} | NT+NN chain hinted {t5 - ln:6.1f} plain {t6 - ln:6.1f}')
