"""Grouped weight gradients (mbv_gemm16_tn_group) against per-layer launches on the multiset of Linear shapes of the
bench step (B = 4, semantic_kitti_512: Swin depths 2-2-6-2, six pixel-decoder layers), MI355X.
python scratch/bench_tn_group.py  ->  microseconds for: per-layer K17 (policy set), per-layer library for the rest,
one grouped call for the policy set, one grouped call for everything.  MBV_GEMM_GROUP_DEPTH sweeps the work-item depth."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mask_bev_amd import ops, tuning  # noqa: E402
from bench_gemm_util import timeit  # noqa: E402

dev = torch.device('cuda', 0)
dt = torch.bfloat16
tuning.use_tuned_gemms()

LAYERS = []      # (tokens, in features, out features)
T, C = 65536, 192
for s, depth in enumerate((2, 2, 6, 2)):
    for _ in range(depth):
        LAYERS += [(T, C, 3 * C), (T, C, C), (T, C, 4 * C), (T, 4 * C, C)]
    if s < 3:
        LAYERS.append((T // 4, 4 * C, 2 * C))
    T //= 4
    C *= 2
LAYERS.append((65536, 2048, 192))                                        # patch projection
for _ in range(6):                                                        # pixel-decoder encoder layers
    LAYERS += [(21504, 256, 544), (21504, 256, 256), (21504, 256, 1024), (21504, 1024, 256)]
LAYERS += [(16384, 256, 768), (4096, 256, 768), (1024, 256, 768)]         # shared key / value projections

items = []
for m, k, n in LAYERS:
    g = torch.randn(m, n, device=dev).to(dt)
    x = torch.randn(m, k, device=dev).to(dt)
    acc = torch.zeros(n, k, device=dev)
    items.append((g, x, acc))


def per_layer_ok(it):
    return it[0].shape[0] >= 4096 and it[1].shape[1] <= 1536


pol = [it for it in items if per_layer_ok(it)]
rest = [it for it in items if not per_layer_ok(it)]
flops = lambda its: sum(2.0 * g.shape[0] * g.shape[1] * x.shape[1] for g, x, _ in its)
byts = lambda its: sum(g.numel() * 2 + x.numel() * 2 + a.numel() * 8 for g, x, a in its)


def run_per_layer(its):
    for g, x, a in its:
        ops.gemm16_tn_acc(a, g, x)


def run_lib(its):
    for g, x, a in its:
        torch.addmm(a, g.t(), x, out=a, out_dtype=torch.float32)


def report(name, us, its):
    print(f'{name:34s} {len(its):3d} products {us:8.1f} us  {flops(its) / us * 1e-6:6.0f} TF/s  {byts(its) / us * 1e-3:6.0f} GB/s')


print('depth', os.environ.get('MBV_GEMM_GROUP_DEPTH', '4096'))
report('per-layer K17 (policy set)', timeit(lambda: run_per_layer(pol), iters=3), pol)
report('grouped K17 (policy set)', timeit(lambda: ops.gemm16_tn_group(sorted(pol, key=lambda it: -it[0].shape[0])), iters=3), pol)
report('per-layer library (rest)', timeit(lambda: run_lib(rest), iters=3), rest)
report('grouped K17 (rest)', timeit(lambda: ops.gemm16_tn_group(sorted(rest, key=lambda it: -it[0].shape[0])), iters=3), rest)
report('grouped K17 (all)', timeit(lambda: ops.gemm16_tn_group(sorted(items, key=lambda it: -it[0].shape[0])), iters=3), items)
