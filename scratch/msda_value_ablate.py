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
for abl in (1, 2, 3, 4):
    os.environ['MBV_MSDA_ABLATE'] = str(abl)
    PART[0] = 1
    bwd(); torch.cuda.synchronize()
    print('plain kernel, ablate', abl, 'value part us:', round(timeit(bwd), 1))
os.environ['MBV_MSDA_ABLATE'] = '0'
PART[0] = 2
print('location/weight part us:', round(timeit(bwd), 1))

# ---- the level launches of the value part (and the location / weight part) on separate streams
os.environ['MBV_MSDA_ABLATE'] = '0'
os.environ['MBV_MSDA_MERGE'] = '0'
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def call(part):
    rc = lib.mbv_ms_deform_attn_bwd(go.data_ptr(), value.data_ptr(), shapes_t.data_ptr(), ls.data_ptr(), loc.data_ptr(),
                                    attn.data_ptr(), B, nv, H, D, L, nq, P, ctypes.cast(host, ctypes.c_void_p),
                                    gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), part, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def serial():
    call(3)


def levels_split(with_loc):
    def f():
        main = torch.cuda.current_stream()
        s1.wait_stream(main)
        if with_loc:
            s2.wait_stream(main)
        with torch.cuda.stream(s1):
            call(1 | (3 << 2))                    # the two coarse levels (8 + 32 KB of LDS per workgroup)
        if with_loc:
            with torch.cuda.stream(s2):
                call(2)                           # d(location), d(weight): no LDS
        call(1 | (4 << 2))                        # the finest level (128 KB)
        if not with_loc:
            call(2)
        main.wait_stream(s1)
        if with_loc:
            main.wait_stream(s2)
    return f


print('whole backward, one stream us:', round(timeit(serial), 1))
print('coarse levels beside the finest us:', round(timeit(levels_split(False)), 1))
print('coarse levels and location / weight part beside the finest us:', round(timeit(levels_split(True)), 1))
