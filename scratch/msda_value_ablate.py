"""K5 backward, value part: where the time goes — full kernel, without the final store, without the LDS atomics, neither
(MBV_MSDA_ABLATE bits 1 / 2; timing experiment only, results are wrong with a bit set)."""
import os, sys, runpy
sys.argv = ['x']
src = open(os.path.join(os.path.dirname(__file__), 'msda_bwd_ab.py')).read().split("for name, env, part in")[0]
exec(src)
for abl in (0, 1, 2, 3):
    os.environ['MBV_MSDA_ABLATE'] = str(abl)
    PART[0] = 1
    bwd(); torch.cuda.synchronize()
    print('ablate', abl, 'value part us:', round(timeit(bwd), 1))
os.environ['MBV_MSDA_ABLATE'] = '0'
PART[0] = 2
print('location/weight part us:', round(timeit(bwd), 1))
