import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
REAL = os.environ.get('K10V', '') == 'real'
lib = ctypes.CDLL(os.environ['K10LIB']) if 'K10LIB' in os.environ else ctypes.CDLL(os.path.join(here, '../../mask_bev_amd/libmaskbev_hip.so') if REAL else os.path.join(here, 'k10s%s.so' % os.environ.get('K10V', '')))
dev = 'cuda'
rows, n, k, H, W = 4000, 37632, 9408, 128, 128
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
g = torch.Generator(device=dev).manual_seed(1)
src = torch.randn((400, H, W), device=dev, generator=g) * scale
if len(sys.argv) > 2 and sys.argv[2] == 'smooth':
    src = torch.nn.functional.interpolate(torch.randn((400, 1, 16, 16), device=dev, generator=g) * scale, size=(H, W), mode='bilinear')[:, 0].contiguous()
idx = (torch.arange(rows, device=dev, dtype=torch.int32) % 400).contiguous()
seed = torch.tensor([1234], device=dev, dtype=torch.int64)
n_rand = 3136
rand = torch.rand((rows, n_rand, 2), device=dev)
out = torch.empty((rows, k + n_rand, 2), device=dev)
P = ctypes.c_void_p
f = lib.mbv_sample_select_uncertain
f.argtypes = [P, P, P, P, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, P, ctypes.c_int32, P, P]
def run():
    rc = f(src.data_ptr(), idx.data_ptr(), None, seed.data_ptr(), rows, n, k, H, W, rand.data_ptr(), n_rand, out.data_ptr(), None)
    assert rc == 0, rc
for _ in range(3): run()
torch.cuda.synchronize()
st = (ctypes.c_ulonglong * 64)()
if not REAL: lib.k10_read_stamps(st, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record(torch.cuda.default_stream())
# the library launches on the null stream (stream = nullptr): time with a device sync instead
import time
t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize()
t1 = time.perf_counter()
if not REAL: lib.k10_read_stamps(st, 1)
print('scale', scale, 'avg ms/launch (host clock, 10 launches)', (t1 - t0) * 100)
tot = sum(st)
for i, v in enumerate(st):
    if v: print('phase', i, 'us per workgroup', round(v / 100 / (10 * rows), 2), 'share', round(v / tot, 3))
print('sum us per workgroup', round(tot / 100 / (10 * rows), 2))
