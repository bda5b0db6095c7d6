"""k_sample_select at the bench shape, 12 calls (for counter collection); logits drawn like trained mask logits."""
import sys
import torch
sys.path.insert(0, '.')
from mask_bev_amd import ops
dev = 'cuda'
R, n, k, H, W, nr = 4000, 37632, 9408, 128, 128, 3136
src = torch.randn(R, H, W, device=dev) * 3.0 - 4.0
idx = torch.arange(R, device=dev, dtype=torch.int32)
rc = torch.rand(R, nr, 2, device=dev)
seed = torch.tensor([12345], dtype=torch.int64, device=dev)
for _ in range(12):
    ops.sample_select_uncertain(src, idx, None, k, rc, seed=seed, num_candidates=n)
torch.cuda.synchronize()
print('ok')
