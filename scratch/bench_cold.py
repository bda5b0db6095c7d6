"""Where do a GEMM's operands have to be for its in-step time?  Stage-2 fused fc1 (16 384 x 384 -> 1 536, bf16, K17) and the library's
plain GEMM of the same shape, graph-timed: (a) the same input every call (warm), (b) the input written by a LayerNorm launch right in
front (the step's situation; the LayerNorm's own time is subtracted), (c) inputs rotating through > 256 MB (cold: beyond the
memory-side cache), (d) weights rotating through > 256 MB, input warm."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scratch._timeit import timeit
from mask_bev_amd import ops
dev = torch.device('cuda', 0)
dt = torch.bfloat16
for (m, k, n) in ((16384, 384, 1536), (4096, 768, 3072), (65536, 192, 768)):
    nbuf = max(2, int(600e6 // (m * k * 2)) )
    xs = [torch.randn(m, k, device=dev).to(dt) for _ in range(min(nbuf, 96))]
    wn = max(2, min(96, int(600e6 // (n * k * 2))))
    ws = [(torch.randn(n, k, device=dev) * 0.05).to(dt) for _ in range(wn)]
    bias = torch.randn(n, device=dev)
    a32 = torch.randn(m, k, device=dev)
    g, b = torch.ones(k, device=dev), torch.zeros(k, device=dev)
    st = dict(i=0)
    def k17(x, w):
        return ops.gemm16_nt(x, w, bias, act='gelu', want_pre=True)
    def lib(x, w):
        return torch.nn.functional.linear(x, w)
    for name, fn in (('k17 fc1+gelu', k17), ('library plain', lib)):
        warm = timeit(lambda: fn(xs[0], ws[0]))
        ln = timeit(lambda: ops.add_layernorm(a32, None, g, b, 1e-5, dt))
        def after_ln():
            y = ops.add_layernorm(a32, None, g, b, 1e-5, dt)
            return fn(y, ws[0])
        both = timeit(after_ln)
        def rot_x():
            st['i'] += 1
            return fn(xs[st['i'] % len(xs)], ws[0])
        cold_x = timeit(rot_x, iters=len(xs))
        def rot_w():
            st['i'] += 1
            return fn(xs[0], ws[st['i'] % len(ws)])
        cold_w = timeit(rot_w, iters=len(ws))
        print(f'{m}x{k}->{n} {name:14s}: warm {warm:6.1f}  behind LayerNorm {both - ln:6.1f} (LN {ln:5.1f})  cold x ({len(xs)} bufs) {cold_x:6.1f}  cold w ({len(ws)}) {cold_w:6.1f}')
