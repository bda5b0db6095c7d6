"""gpurun helper: the grouped column-sum launch on a mix like the step's (a few 44 MB bias gradients of the pixel decoder's FFNs
beside many small partial-row matrices).  python scratch/bench_colsum_group.py [lib.so]"""
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
# the eager step's largest group (scratch/dump_colsum_mix.py): 319 entries, 464 MB, a few of 8-25 MB and a long tail below 1 MB
mats = [torch.randn(65536, 192, device=dev).bfloat16(), torch.randn(16384, 256, device=dev)]
mats += [torch.randn(1024, 6144, device=dev).bfloat16() for _ in range(2)]
mats += [torch.randn(21504, 256, device=dev).bfloat16() for _ in range(6)]
mats += [torch.randn(16384, 256, device=dev).bfloat16() for _ in range(10)]
mats += [torch.randn(1024, 192, device=dev) for _ in range(150)]
mats += [torch.randn(512, 384, device=dev) for _ in range(100)]
mats += [torch.randn(400, 256, device=dev).bfloat16() for _ in range(49)]
outs = [torch.zeros(m.shape[1], device=dev) for m in mats]
n = len(mats)
PA, IA, LA = ctypes.c_void_p * n, ctypes.c_int32 * n, ctypes.c_int64 * n
args = (PA(*[m.data_ptr() for m in mats]), IA(*[ops._dt_flag(m.dtype) for m in mats]), LA(*[m.shape[0] for m in mats]),
        IA(*[m.shape[1] for m in mats]), LA(*[m.shape[1] for m in mats]), PA(*[o.data_ptr() for o in outs]), n)
rc = lib.mbv_colsum_accum_group(*args, ops._stream())
torch.cuda.synchronize()
err = max(float((o - m.float().sum(0)).abs().max() / m.float().sum(0).abs().max()) for o, m in zip(outs, mats))
mb = sum(m.numel() * m.element_size() for m in mats) / 1e6
t = timeit(lambda: lib.mbv_colsum_accum_group(*args, ops._stream()))
print(f'rc {rc} entries {n} {mb:.0f} MB  max rel err {err:.2e}  {t:.1f} us ({mb / t / 1e6 * 1e6 / 1e6:.2f} TB/s)')
