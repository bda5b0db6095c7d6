"""Isolated GPU-bound timings of row-chain programs (100 launches captured in a HIP graph, replayed): the stages of the
decoder's B1 program added one at a time."""
import sys, os, torch
sys.path.insert(0, '.')
from mask_bev_amd import decoder_fused as DF
dev = torch.device('cuda:0')
def r(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)
rows, e, f, wdt = 400, 256, 2048, torch.bfloat16
S = f // 256
x = r((rows, e), 1); w1, b1 = r((f, e), 2, 0.1), r((f,), 3); w2, b2 = r((e, f), 4, 0.05), r((e,), 5)
xb = x.to(wdt)
w1c, w2c = DF.fragment_copy(w1.to(wdt)), DF.fragment_copy(w2.to(wdt))
w2k = DF.fragment_copy(w2.to(wdt), kmajor=True)
wa = DF.fragment_copy(r((e, e), 6, 0.1).to(wdt))
g1, bb = r((e,), 7), r((e,), 8)
hid = torch.empty((rows, f), device=dev); y = torch.empty((rows, e), device=dev); y2 = torch.empty((rows, e), device=dev); y3 = torch.empty((rows, e), device=dev)
parts = torch.empty((S, rows, e), device=dev); stats = torch.empty((rows, 2), device=dev)
def prog(kind):
    split = S if kind.endswith('8') else (3 if kind.endswith('3') else 1)
    P = DF.Program(rows, rows, 1e-5, wdt, split=split)
    k = kind.rstrip('0123456789')
    steps = {'a': 1, 'b': 2, 'c': 3, 'd': 4, 'e': 5, 'f': 6, 'g': 7}.get(k, 0)
    if steps:
        P.load(2, xb, e)
        if steps >= 2:
            with P.only(0):
                P.store(2, y, e)
        if steps >= 3: P.gemm(3, 2, wa, e, e, bias=bb)
        if steps >= 4:
            P.load(4, x, e); P.ln(0, 4, 3, g1, bb, e, stats=stats, save_sum=True)
        if steps >= 5:
            with P.only(0):
                P.store(4, y2, e); P.store(0, y3, e)
        if steps >= 6:
            P.ffn(1, 0 if steps >= 4 else 2, 2 if steps >= 4 else 3, w1c, w2c, e, f, hid, bias_a=b1, bias_out=b2, sliced=split > 1) if False else None
        return P
    if k == 'ffn':
        P.load(0, x, e); P.ffn(1, 0, 2, w1c, w2k if split > 1 else w2c, e, f, hid, bias_a=b1, bias_out=b2, sliced=split > 1)
        if split > 1: P.store_part(1, parts, e)
        else: P.store(1, y, e)
        return P
    if k == 'full':     # the whole B1.fwd
        P.load(2, xb, e)
        with P.only(0):
            P.store(2, y, e)
        P.gemm(3, 2, wa, e, e, bias=bb); P.load(4, x, e); P.ln(0, 4, 3, g1, bb, e, stats=stats, save_sum=True)
        with P.only(0):
            P.store(4, y2, e); P.store(0, y3, e)
        P.ffn(1, 0, 2, w1c, w2k if split > 1 else w2c, e, f, hid, bias_a=b1, bias_out=b2, sliced=split > 1)
        if split > 1: P.store_part(1, parts, e)
        else: P.store(1, y, e)
        return P
    if k == 'sum':
        P.sum_parts(1, parts, e); P.store(1, y, e); return P
    if k == 'gemm':
        P.load(0, x, e)
        P.gemm(1, 0, wa, e, e); P.gemm(2, 1, wa, e, e); P.gemm(1, 2, wa, e, e); P.gemm(2, 1, wa, e, e); P.store(2, y, e); return P
    raise SystemExit(kind)
kinds = sys.argv[1:] or ['a1', 'a8', 'b8', 'c8', 'd8', 'e8', 'ffn1', 'ffn8', 'full1', 'full8', 'sum1', 'gemm1']
for kind in kinds:
    n = 100
    for _ in range(3):
        prog(kind).run()
    torch.cuda.synchronize()
    progs = [prog(kind) for _ in range(n)]
    keep = [list(P.keep) for P in progs]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for P in progs: P.run()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    print(f'{kind:8s} {a.elapsed_time(b) * 1e3 / n:7.2f} us per launch')
