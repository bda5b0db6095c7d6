import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scratch._timeit import timeit
from mask_bev_amd import ops
dev = torch.device('cuda', 0)
for m, n, k in ((65536, 576, 192), (65536, 576, 192), (65536, 192, 192), (65536, 640, 192), (65536, 512, 192), (16384, 1152, 384), (16384, 384, 384), (4096, 2304, 768), (4096, 768, 768), (4096, 768, 3072)):
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05; g = torch.randn(m, n, device=dev)
    am = ops.f32_absmax([x, w]); ag = ops.f32_absmax([g])
    a = timeit(lambda: ops.gemm32s_nt(x, w, None, amax=am))
    b = timeit(lambda: ops.gemm32s_nt(x, w, None, amax=am, hint_out=True))
    c = timeit(lambda: ops.gemm32s_nn(g, w, ag, am[1:2]))
    d = timeit(lambda: ops.gemm32s_nn(g, w, ag, am[1:2], hint_out=True))
    print(f'{m}x{n}x{k}: NT {a:.1f} / hint_out {b:.1f}   NN {c:.1f} / hint_out {d:.1f}')
