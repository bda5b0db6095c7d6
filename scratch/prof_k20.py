"""One K20 shape in a loop, for rocprofv3 --pmc passes: python scratch/prof_k20.py [layout] [m] [n] [k] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mask_bev_amd import ops
layout = sys.argv[1] if len(sys.argv) > 1 else 'nt'
m, n, k = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (16384, 1536, 384)
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device('cuda', 0)
x = torch.randn(m, k, device=dev)
w = torch.randn(n, k, device=dev) * 0.05
g = torch.randn(m, n, device=dev) * 1e-3
acc = torch.zeros(n, k, device=dev)
am, ag = ops.f32_absmax([x, w]), ops.f32_absmax([g])
for _ in range(iters):
    if layout == 'nt':
        ops.gemm32s_nt(x, w, None, amax=am)
    elif layout == 'nn':
        ops.gemm32s_nn(g, w, ag, am[1:2])
    else:
        ops.gemm32s_tn_acc(acc, g, x, ag, am[0:1])
torch.cuda.synchronize()
