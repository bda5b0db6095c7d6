"""gpurun helper: K12 backward (and forward) on the step's shapes, Swin form: dy bf16 + ds f32 -> dx f32 + dx_lo bf16, np = 3, deferred reduce."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mask_bev_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])          # another build of the library (A/B of two kernels)
from mask_bev_amd import ops
from _timeit import timeit
dev = torch.device('cuda', 0)
lib = _lib.load()
P, S = ops._ptr, ops._stream
for rows, c in ((65536, 192), (16384, 384), (4096, 768), (1024, 1536), (21504, 256), (400, 256)):
    dy = torch.randn(rows, c, device=dev).bfloat16(); ds = torch.randn(rows, c, device=dev)
    s = torch.randn(rows, c, device=dev); mean = s.mean(1).contiguous(); rstd = (s.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    gam = torch.rand(c, device=dev) + 0.5; bet = torch.zeros(c, device=dev)
    dx = torch.empty(rows, c, device=dev); dxl = torch.empty(rows, c, device=dev, dtype=torch.bfloat16)
    dg = torch.zeros(c, device=dev); db = torch.zeros(c, device=dev); dbr = torch.zeros(c, device=dev)
    nblk = lib.mbv_add_layernorm_bwd_blocks(rows, c)
    ws = torch.empty(max(1, nblk * 3 * c), device=dev)
    tb = timeit(lambda: lib.mbv_add_layernorm_bwd2(P(dy), 1, None, 0, P(ds), 0, P(s), P(mean), P(rstd), P(gam), rows, c, P(dx), P(dxl), 1,
                                                  P(dg), P(db), 1, P(dbr), P(ws), 1, S()))
    by = rows * c * (2 + 4 + 4 + 4 + 2)
    y = torch.empty(rows, c, device=dev, dtype=torch.bfloat16); so = torch.empty(rows, c, device=dev)
    a = torch.randn(rows, c, device=dev); b = torch.randn(rows, c, device=dev).bfloat16()
    tf = timeit(lambda: lib.mbv_add_layernorm_fwd(P(a), 0, P(b), 1, P(gam), P(bet), rows, c, 1e-5, P(so), P(y), 1, P(mean), P(rstd), S()))
    byf = rows * c * (4 + 2 + 4 + 2)
    torch.cuda.synchronize()
    print(f'rows {rows:6d} C {c:5d}  bwd {tb:6.1f} us ({by / tb / 1e6:5.2f} TB/s, blocks {nblk})   fwd {tf:6.1f} us ({byf / tf / 1e6:5.2f} TB/s)', flush=True)
