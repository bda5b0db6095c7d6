"""gpurun helper: K18's d(added map) kernel (mbv_upsample_bilinear_bwd) on the step's shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mask_bev_amd import ops, _lib
from _timeit import timeit
dev = torch.device('cuda', 0); lib = _lib.load(); P, S = ops._ptr, ops._stream
for dt, flag in ((torch.bfloat16, 1), (torch.float32, 0)):
    gy = torch.randn(4 * 256, 128, 128, device=dev).to(dt); out = torch.empty(4 * 256, 64, 64, device=dev)
    print(dt, '%.1f us' % timeit(lambda: lib.mbv_upsample_bilinear_bwd(P(gy), flag, 1024, 128, 128, 64, 64, P(out), 0, S())))
