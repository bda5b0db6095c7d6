"""K5 backward at the bench shape (B=4, 8 heads x 32, levels 16^2/32^2/64^2, 4 points): the f64-accumulator value part,
the packed fixed-point value part (bf16 output into a 544-wide matrix), the location / weight part.  Graph-timed."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from mask_bev_amd import _lib, ops
lib = _lib.load()
dev = torch.device('cuda:0')
B, H, D, P = 4, 8, 32, 4
shapes = [(16, 16), (32, 32), (64, 64)]
L = len(shapes); nv = sum(h * w for h, w in shapes)
g = torch.Generator(device=dev).manual_seed(0)
refs = []
for (h, w) in shapes:
    xs = (torch.arange(w, device=dev) + 0.5) / w; ys = (torch.arange(h, device=dev) + 0.5) / h
    refs.append(torch.stack([xs.repeat(h), ys.view(-1, 1).repeat(1, w).view(-1)], -1))
ref = torch.cat(refs, 0)
norm = torch.tensor([[w, h] for h, w in shapes], device=dev, dtype=torch.float32)
spread = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
off = torch.randn(B, nv, H, L, P, 2, device=dev, generator=g) * spread
loc = (ref.view(1, nv, 1, 1, 1, 2) + off / norm.view(1, 1, 1, L, 1, 2)).contiguous()
attn = torch.rand(B, nv, H, L, P, device=dev, generator=g).flatten(-2).softmax(-1).view(B, nv, H, L, P).contiguous()
value = torch.randn(B, nv, H, D, device=dev, generator=g)
go = torch.randn(B, nv, H * D, device=dev, generator=g)
shapes_t = torch.tensor(shapes, dtype=torch.int64, device=dev)
starts = [0]
for h, w in shapes[:-1]: starts.append(starts[-1] + h * w)
ls = torch.tensor(starts, dtype=torch.int64, device=dev)
host = (ctypes.c_int64 * (2 * L))(*[int(v) for hw in shapes for v in hw])
gv = torch.empty_like(value); gl = torch.empty_like(loc); ga = torch.empty_like(attn)
G = torch.empty(B * nv, 544, dtype=torch.bfloat16, device=dev)
P_ = ops._ptr
WS = torch.empty(lib.mbv_ms_deform_attn_bwd_value_packed_workspace_bytes(B, H, L, nv), dtype=torch.uint8, device=dev)

def f64_value():
    ops.check(lib.mbv_ms_deform_attn_bwd(P_(go), P_(value), P_(shapes_t), P_(ls), P_(loc), P_(attn), B, nv, H, D, L, nv, P, host,
                                         P_(gv), P_(None), P_(None), 1, ops._stream()), 'bwd')
def locattn():
    ops.check(lib.mbv_ms_deform_attn_bwd(P_(go), P_(value), P_(shapes_t), P_(ls), P_(loc), P_(attn), B, nv, H, D, L, nv, P, host,
                                         P_(None), P_(gl), P_(ga), 2, ops._stream()), 'bwd')
def packed():
    ops.check(lib.mbv_ms_deform_attn_bwd_value_packed(P_(go), P_(loc), P_(attn), B, nv, H, D, L, nv, P, host, P_(G), 1, 544,
                                                      P_(WS), WS.numel(), ops._stream()), 'packed')
def packed_f32():
    ops.check(lib.mbv_ms_deform_attn_bwd_value_packed(P_(go), P_(loc), P_(attn), B, nv, H, D, L, nv, P, host, P_(gv), 0, 256,
                                                      P_(WS), WS.numel(), ops._stream()), 'packed')
def cast():
    G[:, :256].copy_(gv.view(-1, 256))

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr.replay(); torch.cuda.synchronize()
    a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

f64_value(); refv = gv.clone(); packed_f32(); torch.cuda.synchronize()
print('packed(f32 out) vs f64 max abs diff', float((gv - refv).abs().max()), 'max|ref|', float(refv.abs().max()))
print(f'offset spread {spread} px')
for name, fn in [('value f64 (3 launches)', f64_value), ('value packed -> bf16 ld 544', packed), ('value packed -> f32', packed_f32),
                 ('f32 -> bf16 cast into G', cast), ('location / weight part', locattn)]:
    print(f'{name:32s} {timeit(fn):8.1f} us')
