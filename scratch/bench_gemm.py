"""K17 vs the library GEMM on the Linear shapes of the bench step (B = 4, semantic_kitti_512), MI355X.
python scratch/bench_gemm.py  ->  one line per (layer, pass): microseconds for torch (hipBLASLt) and K17."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mask_bev_amd import ops, tuning  # noqa: E402

dev = torch.device('cuda', 0)
dt = torch.bfloat16
tuning.use_tuned_gemms()


def timeit(fn, iters=20, warm=3, reps=5):
    """Device time per call: `iters` calls captured in one HIP graph (no host launch gaps), best of `reps` replays."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


SHAPES = []      # (name, M tokens, in features, out features)
T, C = 65536, 192
for s in range(4):
    SHAPES += [(f's{s + 1}.qkv', T, C, 3 * C), (f's{s + 1}.proj', T, C, C), (f's{s + 1}.fc1', T, C, 4 * C),
               (f's{s + 1}.fc2', T, 4 * C, C)]
    if s < 3:
        SHAPES.append((f's{s + 1}.merge', T // 4, 4 * C, 2 * C))
    T //= 4
    C *= 2
SHAPES += [('patch', 65536, 2048, 192), ('pd.value', 21504, 256, 256), ('pd.ffn1', 21504, 256, 1024),
           ('pd.ffn2', 21504, 1024, 256), ('pd.qside', 21504, 256, 544), ('kv.l0', 1024, 256, 768),
           ('kv.l2', 16384, 256, 768)]

tot = {'fwd': [0, 0], 'dgrad': [0, 0], 'wgrad': [0, 0]}
print(f'{"layer":10s} {"M":>6s} {"K":>5s} {"N":>5s} | fwd lib / k17 (TF/s) | dgrad lib / k17 | wgrad lib / k17   [us]')
for name, m, k, n in SHAPES:
    x = torch.randn(m, k, device=dev).to(dt)
    w = (torch.randn(n, k, device=dev) * 0.05).to(dt)
    g = torch.randn(m, n, device=dev).to(dt)
    b32 = torch.randn(n, device=dev)
    bb = b32.to(dt)
    acc = torch.zeros(n, k, device=dev)
    fl = 2.0 * m * n * k
    t_f_lib = timeit(lambda: torch.nn.functional.linear(x, w, bb))
    t_f_k = timeit(lambda: ops.gemm16_nt(x, w, b32))
    t_d_lib = timeit(lambda: g.mm(w))
    t_d_k = timeit(lambda: ops.gemm16_nn(g, w))
    os.environ['MBV_GEMM16'] = '0'
    t_w_lib = timeit(lambda: ops._wgrad_into(acc, g, x))          # library path: batched split-K GEMM + column sums
    os.environ['MBV_GEMM16'] = 'auto'
    t_w_k = timeit(lambda: ops.gemm16_tn_acc(acc, g, x))
    for key, a, b in (('fwd', t_f_lib, t_f_k), ('dgrad', t_d_lib, t_d_k), ('wgrad', t_w_lib, t_w_k)):
        tot[key][0] += a
        tot[key][1] += b
    print(f'{name:10s} {m:6d} {k:5d} {n:5d} | {t_f_lib:7.1f} / {t_f_k:7.1f} ({fl / t_f_k / 1e6:5.0f}) | '
          f'{t_d_lib:7.1f} / {t_d_k:7.1f} ({fl / t_d_k / 1e6:5.0f}) | {t_w_lib:7.1f} / {t_w_k:7.1f} ({fl / t_w_k / 1e6:5.0f})',
          flush=True)
print('totals [us]:', {k: (round(v[0]), round(v[1])) for k, v in tot.items()})
# fused epilogues
m, c = 65536, 192
x = torch.randn(m, c, device=dev).to(dt)
w1 = (torch.randn(4 * c, c, device=dev) * 0.05).to(dt)
b1 = torch.randn(4 * c, device=dev)
b1h = b1.to(dt)
print('fc1+gelu lib (linear + gelu):', timeit(lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, w1, b1h))),
      'k17 fused (out + pre):', timeit(lambda: ops.gemm16_nt(x, w1, b1, act='gelu', want_pre=True)))
g = torch.randn(m, c, device=dev).to(dt)
w2 = (torch.randn(c, 4 * c, device=dev) * 0.05).to(dt)
h = torch.randn(m, 4 * c, device=dev).to(dt)
cs = torch.zeros(4 * c, device=dev)
print('fc2 dgrad + gelu\' + colsum k17 fused:', timeit(lambda: ops.gemm16_nn(g, w2, act='gelu', aux=h, colsum=cs)),
      ' unfused k17 dgrad alone:', timeit(lambda: ops.gemm16_nn(g, w2)))
