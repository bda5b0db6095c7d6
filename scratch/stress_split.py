"""Stress: the split-launch MLP of the row chain, repeated; every repetition must equal the first bit for bit."""
import sys, torch
sys.path.insert(0, '.')
from mask_bev_amd import decoder_fused as DF
dev = torch.device('cuda:0')
def r(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)
rows, e, f, wdt = 400, 256, 2048, torch.bfloat16
S = f // 256
x = r((rows, e), 1); w1, b1 = r((f, e), 2, 0.1), r((f,), 3); w2, b2 = r((e, f), 4, 0.05), r((e,), 5)
w1c, w2c = DF.fragment_copy(w1.to(wdt)), DF.fragment_copy(w2.to(wdt))
w2k = DF.fragment_copy(w2.to(wdt), kmajor=True)
wa = DF.fragment_copy(r((e, e), 6, 0.1).to(wdt))
def once():
    hid = torch.empty((rows, f), device=dev); y = torch.empty((rows, e), device=dev); ga = torch.empty((rows, e), device=dev)
    parts = torch.empty((S, rows, e), device=dev)
    P = DF.Program(rows, rows, 1e-5, wdt, split=S)
    P.load(1, x, e)
    P.gemm(0, 1, wa, e, e)
    with P.only(0):
        P.store(0, ga, e)
    P.ffn(1, 0, 2, w1c, w2k, e, f, hid, bias_a=b1, bias_out=b2, sliced=True)
    P.store_part(1, parts, e)
    P.run()
    P = DF.Program(rows, rows, 1e-5, wdt, split=3)
    P.sum_parts(1, parts, e)
    with P.only(1):
        P.gemm(2, 1, wa, e, e, out=y)
    P.run()
    return hid, y, ga
ref = once(); torch.cuda.synchronize()
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for i in range(n):
    got = once()
    if i % 50 == 49:
        torch.cuda.synchronize()
    if not all(torch.equal(a, b) for a, b in zip(ref, got)):
        bad += 1
        print('mismatch at', i, [float((a - b).abs().max()) for a, b in zip(ref, got)])
        if bad > 5: break
torch.cuda.synchronize()
print('stress done', n, 'bad', bad)
