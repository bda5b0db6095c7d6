"""K5 backward, value part: (a) the neighbour-merging kernel vs the plain one, for random 2-px offsets and for the
module's initial offset pattern (identical offsets for all queries); (b) where the plain kernel's time goes — without the
final store, without the LDS atomics, neither (MBV_MSDA_ABLATE bits 1 / 2; timing only, results are wrong with a bit set)."""
import os, sys
sys.argv = ['x']
src = open(os.path.join(os.path.dirname(__file__), 'msda_bwd_ab.py')).read().split("for name, env, part in")[0]
exec(src)
for pattern in ('random 2-px offsets', 'initial pattern (same offsets for every query)'):
    if pattern.startswith('initial'):
        off0 = torch.randn(1, 1, H, L, P, 2, device=dev, generator=g) * 2.0
        loc.copy_(ref.view(1, nq, 1, 1, 1, 2) + off0 / norm.view(1, 1, 1, L, 1, 2))
    for merge in ('0', '1'):
        os.environ['MBV_MSDA_MERGE'] = merge
        os.environ['MBV_MSDA_ABLATE'] = '0'
        PART[0] = 1
        bwd(); torch.cuda.synchronize()
        print(f'{pattern}: merge={merge} value part us: {timeit(bwd):.1f}  checksum {float(gv.double().abs().sum()):.6e}')
os.environ['MBV_MSDA_MERGE'] = '0'
for abl in (1, 2, 3):
    os.environ['MBV_MSDA_ABLATE'] = str(abl)
    PART[0] = 1
    bwd(); torch.cuda.synchronize()
    print('plain kernel, ablate', abl, 'value part us:', round(timeit(bwd), 1))
os.environ['MBV_MSDA_ABLATE'] = '0'
PART[0] = 2
print('location/weight part us:', round(timeit(bwd), 1))
