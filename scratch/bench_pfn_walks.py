"""gpurun helper: the PFN's per-pillar walk kernels on synthetic pillars of the bench batch's size (K = 440 668 rows in
V = 83 722 pillars, 1..32 rows each, mean 5.3) — time per launch and a checksum of every output (to compare two builds).
python scratch/bench_pfn_walks.py [lib.so]"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mask_bev_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from mask_bev_amd import ops
from _timeit import timeit
lib = _lib.load()
dev = torch.device('cuda', 0)
torch.manual_seed(0)
V, P = 83722, 32
n = torch.clamp((torch.rand(V) ** 3 * 20).long() + 1, max=P)
n = (n.float() * (440668 / n.sum().item())).round().long().clamp(1, P)
K = int(n.sum())
row_start = torch.cat([torch.zeros(1, dtype=torch.long), n.cumsum(0)[:-1]]).int().to(dev)
num_points = n.int().to(dev)
print('rows', K, 'pillars', V)
Pt, S = ops._ptr, ops._stream
for U in (64, 128):
    y = torch.randn(K, U, device=dev); ypad = torch.randn(V, U, device=dev)
    dz0 = torch.randn(K, U, device=dev); dzpad0 = torch.randn(V, U, device=dev)
    mean = torch.randn(U, device=dev) * 0.1; rstd = torch.rand(U, device=dev) + 0.5; gamma = torch.rand(U, device=dev) + 0.5
    sums = (torch.randn(2 * U, device=dev, dtype=torch.float64) * 1000)
    dt = torch.empty(V, U, device=dev)
    dz, dzpad = dz0.clone(), dzpad0.clone()
    call = lambda: lib.mbv_pfn_bwd_bn(Pt(y), Pt(ypad), Pt(dz), Pt(dzpad), Pt(mean), Pt(rstd), Pt(gamma), Pt(sums),
                                      ctypes.c_double(float(V) * P), 1, Pt(row_start), Pt(num_points), V, U, P, Pt(dt), S())
    rc = call(); torch.cuda.synchronize()
    print(f'U={U} bwd_bn rc={rc} checksums dz {dz.double().sum().item():.6e} |dz| {dz.double().abs().sum().item():.6e} '
          f'dzpad {dzpad.double().sum().item():.6e} dt {dt.double().sum().item():.6e} |dt| {dt.double().abs().sum().item():.6e}')
    print(f'U={U} bwd_bn  %.1f us' % timeit(call))
    # apply + max
    scale = torch.rand(U, device=dev) + 0.5; shift = torch.randn(U, device=dev) * 0.3
    a = torch.empty(K, U, device=dev); apad = torch.empty(V, U, device=dev); m = torch.empty(V, U, device=dev)
    call = lambda: lib.mbv_pfn_apply_max(Pt(y), Pt(ypad), Pt(scale), Pt(shift), Pt(row_start), Pt(num_points), V, U, P,
                                         Pt(a), Pt(apad), Pt(m), S())
    rc = call(); torch.cuda.synchronize()
    print(f'U={U} apply_max rc={rc} checksums a {a.double().sum().item():.9e} apad {apad.double().sum().item():.9e} m {m.double().sum().item():.9e}')
    print(f'U={U} apply_max  %.1f us' % timeit(call))
    # route
    dm = torch.randn(V, U, device=dev); sapad = torch.randn(V, U, device=dev)
    dzr = dz0.clone(); dzpad = torch.empty(V, U, device=dev); sums2 = torch.zeros(2 * U, device=dev, dtype=torch.float64)
    call = lambda: lib.mbv_pfn_bwd_route(Pt(y), Pt(ypad), Pt(scale), Pt(shift), Pt(mean), Pt(rstd), Pt(dzr), 1, Pt(sapad), Pt(dm),
                                         Pt(row_start), Pt(num_points), V, U, P, Pt(dzpad), Pt(sums2), S())
    rc = call(); torch.cuda.synchronize()
    print(f'U={U} bwd_route rc={rc} checksums dz {dzr.double().sum().item():.9e} |dz| {dzr.double().abs().sum().item():.9e} '
          f'dzpad {dzpad.double().sum().item():.9e} sums {sums2.sum().item():.9e} |sums| {sums2.abs().sum().item():.9e}')
    print(f'U={U} bwd_route  %.1f us' % timeit(call))
