"""K5 backward: 4-channels-per-lane banded form vs the one-channel form (MBV_MSDA_BWD_1CH=1), device time of the
backward kernel alone (HIP-graph replays), bench shapes.  Offsets: the module's initial bias pattern (1..4 px)."""
import os, sys, torch, ctypes
sys.path.insert(0, '.')
from mask_bev_amd import ops, _lib
dev = torch.device('cuda:0')
B, H, D, P = 4, 8, 32, 4
shapes = [(16, 16), (32, 32), (64, 64)]
L = len(shapes)
nv = sum(h * w for h, w in shapes); nq = nv
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(B, nv, H, D, device=dev, generator=g)
refs = []
for (h, w) in shapes:
    xs = (torch.arange(w, device=dev) + 0.5) / w; ys = (torch.arange(h, device=dev) + 0.5) / h
    refs.append(torch.stack([xs.repeat(h), ys.view(-1, 1).repeat(1, w).view(-1)], -1))
ref = torch.cat(refs, 0)
norm = torch.tensor([[w, h] for h, w in shapes], device=dev, dtype=torch.float32)
off = torch.randn(B, nq, H, L, P, 2, device=dev, generator=g) * 2.0
loc = (ref.view(1, nq, 1, 1, 1, 2) + off / norm.view(1, 1, 1, L, 1, 2)).contiguous()
attn = torch.rand(B, nq, H, L, P, device=dev, generator=g).flatten(-2).softmax(-1).view(B, nq, H, L, P).contiguous()
shapes_t = torch.tensor(shapes, dtype=torch.int64, device=dev)
starts = [0]
for h, w in shapes[:-1]: starts.append(starts[-1] + h * w)
ls = torch.tensor(starts, dtype=torch.int64, device=dev)
go = torch.randn(B, nq, H * D, device=dev, generator=g)
lib = _lib.load()
host = (ctypes.c_int64 * (2 * L))(*[v for hw in shapes for v in hw])
gv, gl, ga = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(attn)

PART = [3]


def bwd():
    rc = lib.mbv_ms_deform_attn_bwd(go.data_ptr(), value.data_ptr(), shapes_t.data_ptr(), ls.data_ptr(), loc.data_ptr(),
                                    attn.data_ptr(), B, nv, H, D, L, nq, P, ctypes.cast(host, ctypes.c_void_p),
                                    gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), PART[0], torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc

def timeit(fn, iters=10, reps=5):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters): fn()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best

for name, env, part in (('no-atomics form (both parts, one stream)', {}, 3), ('  value part only', {}, 1),
                        ('  location/weight part only', {}, 2), ('banded form', {'MBV_MSDA_BWD_BANDED': '1'}, 3)):
    os.environ.pop('MBV_MSDA_BWD_BANDED', None)
    os.environ.update(env)
    PART[0] = part
    bwd(); torch.cuda.synchronize()
    if not env or 'BANDED' in str(env):
        print(name, 'us:', timeit(bwd), ' grad_value checksum', float(gv.double().abs().sum()), float(gl.double().abs().sum()))
    else:
        print(name, 'us:', timeit(bwd))
