"""K20 (f32 products from IEEE-half pairs, csrc/gemm_f32s.hip) against the library's f32 GEMM on the fp32 step's token-major
Linear shapes (B = 4, semantic_kitti_512): forward NT, data gradient NN, weight gradient TN, + the absmax pass K20 needs.
Graph-replayed device time (scratch/_timeit.py).  python scratch/bench_gemm32s.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scratch._timeit import timeit
from mask_bev_amd import ops

dev = torch.device('cuda', 0)
shapes = []
for tokens, c in ((65536, 192), (16384, 384), (4096, 768), (1024, 1536)):
    shapes += [(f'qkv  T={tokens}', tokens, 3 * c, c), (f'proj T={tokens}', tokens, c, c),
               (f'fc1  T={tokens}', tokens, 4 * c, c), (f'fc2  T={tokens}', tokens, c, 4 * c)]
shapes += [('pixdec ffn1 T=21504', 21504, 1024, 256), ('pixdec ffn2 T=21504', 21504, 256, 1024),
           ('pixdec proj T=21504', 21504, 256, 256)]
tot = dict(lib_nt=0, k_nt=0, lib_nn=0, k_nn=0, lib_tn=0, k_tn=0, amax=0)
print(f'{"shape":24s} {"GF":>6s} | NT lib   K20  | NN lib   K20  | TN lib   K20  | absmax(x,w) (us)')
for name, m, n, k in shapes:
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * 0.05
    g = torch.randn(m, n, device=dev) * 1e-3
    acc = torch.zeros(n, k, device=dev)
    bias = torch.randn(n, device=dev)
    am = ops.f32_absmax([x, w])
    ag = ops.f32_absmax([g])
    t = {}
    t['lib_nt'] = timeit(lambda: torch.addmm(bias, x, w.t()))
    t['k_nt'] = timeit(lambda: ops.gemm32s_nt(x, w, bias, amax=am))
    t['lib_nn'] = timeit(lambda: g @ w)
    t['k_nn'] = timeit(lambda: ops.gemm32s_nn(g, w, ag, am[1:2]))
    t['lib_tn'] = timeit(lambda: ops._wgrad_into(acc, g, x))
    t['k_tn'] = timeit(lambda: ops.gemm32s_tn_acc(acc, g, x, ag, am[0:1]))
    t['amax'] = timeit(lambda: ops.f32_absmax([x, w]))
    for kk in tot:
        tot[kk] += t[kk]
    gf = 2.0 * m * n * k / 1e9
    print(f'{name:24s} {gf:6.1f} | {t["lib_nt"]:6.1f} {t["k_nt"]:6.1f} | {t["lib_nn"]:6.1f} {t["k_nn"]:6.1f} | '
          f'{t["lib_tn"]:6.1f} {t["k_tn"]:6.1f} | {t["amax"]:6.1f}   K20 NT {gf / t["k_nt"] * 1e-3:6.1f} TF-equivalent')
print('sum', {k: round(v, 1) for k, v in tot.items()})
