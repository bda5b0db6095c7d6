"""One K17 weight-gradient shape, 30 calls (for counter collection)."""
import sys
import torch
sys.path.insert(0, '.')
from mask_bev_amd import ops
dev = torch.device('cuda', 0)
M, K, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (65536, 192, 768)
g = torch.randn(M, N, device=dev).bfloat16()
x = torch.randn(M, K, device=dev).bfloat16()
acc = torch.zeros(N, K, device=dev)
for _ in range(30):
    ops.gemm16_tn_acc(acc, g, x)
torch.cuda.synchronize()
print('ok', float(acc.abs().mean()))
