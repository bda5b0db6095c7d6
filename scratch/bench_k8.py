"""gpurun helper: K8 at the bench step's shapes — forward from the LDS map (4 000 rows x 12 544 points, 128 x 128 logits),
forward from bit-packed 512 x 512 targets, backward (stack form, bf16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mask_bev_amd import ops, _lib
from _timeit import timeit
dev = torch.device('cuda', 0)
lib = _lib.load()
d, b, q, h, w, p = 10, 4, 100, 128, 128, 12544
n = d * b * q
torch.manual_seed(0)
idx = torch.arange(n, dtype=torch.int32, device=dev)
src = torch.randn(n, h, w, device=dev)
coords = torch.rand(n, p, 2, device=dev)             # per-row coordinates (the loss's selected points)
out = torch.empty(n, p, device=dev)
P, S = ops._ptr, ops._stream
print('fwd lds, per-row coords   %.1f us' % timeit(lambda: lib.mbv_point_sample_fwd(P(src), P(idx), P(coords), P(idx), n, p, h, w, P(out), S())))
cshared = torch.rand(d * b, p, 2, device=dev)        # the matcher's points: one set per (decoder output, image)
cidx = (idx // q).to(torch.int32)
print('fwd lds, shared coords    %.1f us' % timeit(lambda: lib.mbv_point_sample_fwd(P(src), P(idx), P(cshared), P(cidx), n, p, h, w, P(out), S())))
gt = (torch.rand(b * q, 512, 512, device=dev) > 0.5).float()
packed = ops.pack_binary_masks(gt)
gidx = (idx % (b * q)).to(torch.int32)
print('fwd packed 512x512        %.1f us' % timeit(lambda: lib.mbv_point_sample_packed_fwd(P(packed.words), P(gidx), P(coords), P(idx), n, p, 512, 512, P(out), S())))
gout = torch.randn(n, p, device=dev)
stack = torch.empty(b, d, q, h * w, dtype=torch.bfloat16, device=dev)
print('bwd stack bf16            %.1f us' % timeit(lambda: lib.mbv_point_sample_bwd_stack(P(gout), P(idx), P(coords), P(idx), n, p, h, w, d, b, q, P(stack), ops._dt_flag(torch.bfloat16), S())))
# reference check of the forward against F.grid_sample on a few rows
import torch.nn.functional as F
lib.mbv_point_sample_fwd(P(src), P(idx), P(coords), P(idx), n, p, h, w, P(out), S())
ref = F.grid_sample(src[:8, None], coords[:8, :, None, :] * 2 - 1, align_corners=False)[:, 0, :, 0]
print('max |fwd - grid_sample| on 8 rows: %.3g' % (out[:8] - ref).abs().max().item())
