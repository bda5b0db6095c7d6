"""Ablation of K17's products (mbv_gemm16_nt / _nn): what each part of the kernel costs on the step's shapes, and the
experimental 256 x 192 block shape (12 waves, one workgroup per CU, fragments of step kt + 1 read under step kt's multiplies;
gemm_shape2.patch — measured, not adopted: results in scratch/ubench/README.md).
`python scratch/ubench/gemm_abl.py build` (CPU container: hipcc cross-compiles the variants into scratch/ubench/*.so),
`python scratch/ubench/gemm_abl.py` on the GPU box times them.  Variants are made by patching a copy of csrc/gemm.hip:
  bit 1  no global stores in the epilogue        bit 2  every LDS-DMA piece is a zero fill (no memory traffic)
  bit 4  no matrix instructions                  bit 8  no fragment reads from LDS
  bit 16 plain blockIdx work order"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'mask_bev_amd', 'csrc')
VARIANTS = [(2, v, 4) for v in (0, 1, 2, 4, 8, 3, 5, 9, 6, 10, 12, 14, 13, 11, 7, 15)]      # (forced block shape, ablation bits, ring slots of shape 2)


def patched_source():
    # csrc/gemm.hip + gemm_shape2.patch (the experimental 256 x 192 shape with the pipelined K loop: not in the product)
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, 'mask_bev_amd', 'csrc'))
        dst = os.path.join(tmp, 'mask_bev_amd', 'csrc', 'gemm.hip')
        shutil.copy(os.path.join(CSRC, 'gemm.hip'), dst)
        subprocess.check_call(['patch', '-s', '-p1', '-d', tmp, '-i', os.path.join(HERE, 'gemm_shape2.patch')])
        s = open(dst).read()

    def rep(old, new, count=1):
        nonlocal s
        assert s.count(old) >= 1, old
        s = s.replace(old, new, count)

    rep('#include "common.hpp"\n#include "gemm_tiles.hpp"',
        f'#include "{CSRC}/common.hpp"\n#include "{CSRC}/gemm_tiles.hpp"')
    rep('      store8<T>(p.c, o, v, p.out_f32);\n',
        '      if (!(MBV_ABL & 1) || v[0] == 123456.f) store8<T>(p.c, o, v, p.out_f32);\n')
    rep('      if (kt >= nk) { v = OOB; sof = 0u; }',
        '      if ((MBV_ABL & 2) || kt >= nk) { v = OOB; sof = 0u; }')
    rep('        af[ks][i] = A_KS ? frag_ks(im, x & 127, ks, lane) : frag_kc<KB>(im, x & 127, ks, lane);',
        '        if (MBV_ABL & 8) af[ks][i] = make_uint4(lane, ks, i, kt); else\n'
        '        af[ks][i] = A_KS ? frag_ks(im, x & 127, ks, lane) : frag_kc<KB>(im, x & 127, ks, lane);')
    rep('        bf[ks][j] = B_KS ? frag_ks(im, x & 127, ks, lane) : frag_kc<KB>(im, x & 127, ks, lane);',
        '        if (MBV_ABL & 8) bf[ks][j] = make_uint4(lane, ks, j, kt); else\n'
        '        bf[ks][j] = B_KS ? frag_ks(im, x & 127, ks, lane) : frag_kc<KB>(im, x & 127, ks, lane);')
    rep('          acc[i][j] = OUT == 0 ? Mma<T>::run(bf[ks][j], af[ks][i], acc[i][j]) : Mma<T>::run(af[ks][i], bf[ks][j], acc[i][j]);',
        '          if (MBV_ABL & 4) acc[i][j][ks] += __uint_as_float(af[ks][i].x ^ bf[ks][j].y); else\n'
        '          acc[i][j] = OUT == 0 ? Mma<T>::run(bf[ks][j], af[ks][i], acc[i][j]) : Mma<T>::run(af[ks][i], bf[ks][j], acc[i][j]);')
    rep('gemm16_body<KB, NS, WM, TM, TN, A_KS, B_KS, OUT, EPI, T, WN>(p, xcd_contiguous(blockIdx.x, gridDim.x), smem);',
        'gemm16_body<KB, NS, WM, TM, TN, A_KS, B_KS, OUT, EPI, T, WN>(p, (MBV_ABL & 16) ? (int)blockIdx.x : xcd_contiguous(blockIdx.x, gridDim.x), smem);')
    rep('constexpr int NS2 = 4;', 'constexpr int NS2 = MBV_NS2;')
    # in-kernel stamps (diagnostic build only): wave 0 of every workgroup, s_memtime at entry / first fragments / end of
    # the K loop / end of the epilogue, and the K loop's time split into "wait + barrier" and "issue + multiply"
    rep('namespace {\n\n// Block shape:',
        'namespace {\n__device__ unsigned long long g_stamps[8 * 8192];\n'
        '#define MBV_ST(x) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(x) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)\n\n// Block shape:')
    rep('  const int wm = wave / WN, wn = wave - WN * wm;\n',
        '  const int wm = wave / WN, wn = wave - WN * wm;\n'
        '  unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, stA = 0, stB = 0, st_wait = 0, st_work = 0;\n'
        '  if (MBV_STAMP) MBV_ST(st0);\n')
    rep('    read_frags(0, af0, bf0);\n', '    read_frags(0, af0, bf0);\n    if (MBV_STAMP) { MBV_ST(st1); stB = st1; }\n')
    for tail in ('stage(kt + NS - 1);', 'stage(kt + NS);'):
        rep('      wait_pieces(std::integral_constant<int, NS - 3>());\n      __syncthreads();\n      ' + tail,
            '      if (MBV_STAMP) { MBV_ST(stA); st_work += stA - stB; }\n'
            '      wait_pieces(std::integral_constant<int, NS - 3>());\n      __syncthreads();\n'
            '      if (MBV_STAMP) { MBV_ST(stB); st_wait += stB - stA; }\n      ' + tail)
    rep('  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the zero fills of the tail',
        '  if (MBV_STAMP) MBV_ST(st2);\n  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the zero fills of the tail')
    rep('  if (p.colsum_rows) {\n#pragma unroll\n    for (int e = 0; e < 8; ++e) {\n      float s = csum[e];',
        '  if (MBV_STAMP) {\n    MBV_ST(st3);\n    if (tid == 0 && bid < 8192) {\n      unsigned long long* d = g_stamps + 8 * bid;\n'
        '      d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3; d[4] = st_wait; d[5] = st_work; d[6] = nk;\n'
        '      unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); d[7] = x;\n    }\n  }\n'
        '  if (p.colsum_rows) {\n#pragma unroll\n    for (int e = 0; e < 8; ++e) {\n      float s = csum[e];')
    s += ('\nextern "C" int abl_read_stamps(unsigned long long* host, int n) {\n'
          '  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), (size_t)n * 8);\n}\n')
    rep('static int gemm16_pick_shape(int atomic, long long gm, long long gn, long long work_units) {',
        'static int gemm16_pick_shape(int atomic, long long gm, long long gn, long long work_units) {\n  if (!atomic) return MBV_FORCE_SHAPE;')
    return s


def build():
    src = os.path.join(HERE, '_gemm_abl_src.hip')
    open(src, 'w').write(patched_source())
    for sh, v, ns2 in VARIANTS:
        out = os.path.join(HERE, f'libgemm_abl_{sh}_{v}_{ns2}.so')
        cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-fast-math', '-shared',
               f'-DMBV_ABL={v & 255}', f'-DMBV_STAMP={v >> 8}', f'-DMBV_FORCE_SHAPE={sh}', f'-DMBV_NS2={ns2}', f'-I{CSRC}', f'-I{ROOT}/include', src, '-o', out]
        print(' '.join(cmd[-4:]), flush=True)
        subprocess.check_call(cmd)
    os.remove(src)


def run():
    import torch
    sys.path.insert(0, ROOT)
    from scratch._timeit import timeit        # graph-timed, best of the replays
    dev = torch.device('cuda', 0)
    shapes = [('s3.fc1', 4096, 768, 3072), ('s1.fc2', 65536, 768, 192)]
    nt, nn, stamp_readers = {}, {}, {}
    for key in VARIANTS:
        lib = ctypes.CDLL(os.path.join(HERE, f'libgemm_abl_{key[0]}_{key[1]}_{key[2]}.so'))
        f = lib.mbv_gemm16_nt
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64] * 6 + [ctypes.c_int32] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_void_p]
        nt[key] = f
        f = lib.mbv_gemm16_nn
        f.restype = ctypes.c_int
        f.argtypes = ([ctypes.c_void_p] * 5 + [ctypes.c_int64] * 7 + [ctypes.c_int32] * 4 + [ctypes.c_int64] * 3 +
                      [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p])
        nn[key] = f
        if key[1] >> 8:
            f = lib.abl_read_stamps
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.c_void_p, ctypes.c_int]
            stamp_readers[key] = f
    print(f'{"shape":8s} {"pass":5s} {"lib":>7s} ' + ' '.join(f'a{b}'.rjust(7) for a, b, c in VARIANTS) + '   [us]')
    for name, m, k, n in shapes:
        x = torch.randn(m, k, device=dev).bfloat16()
        w = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
        g = torch.randn(m, n, device=dev).bfloat16()
        b = torch.randn(n, device=dev)
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        dx = torch.empty(m, k, device=dev, dtype=torch.bfloat16)
        bb = b.bfloat16()
        st = torch.cuda.current_stream().cuda_stream
        ref = torch.nn.functional.linear(x.float(), w.float(), b)
        ref_dx = g.float() @ w.float()
        t_lib = timeit(lambda: torch.nn.functional.linear(x, w, bb))
        t_lib_nn = timeit(lambda: g.mm(w))
        row, row_nn = [], []
        for key in VARIANTS:
            def call(f=nt[key]):
                rc = f(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), None, m, n, k, k, k, n, 0, 0, 0, 1, 0, 0, 0,
                       torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc

            def call_nn(f=nn[key]):
                rc = f(g.data_ptr(), w.data_ptr(), dx.data_ptr(), None, None, m, n, k, n, k, k, 0, 0, 0, 0, 1, 0, 0, 0, None, 0,
                       torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            if key[1] == 0:
                out.zero_(); dx.zero_()
                call(); call_nn()
                e1 = ((out.float() - ref).abs().max() / ref.abs().max()).item()
                e2 = ((dx.float() - ref_dx).abs().max() / ref_dx.abs().max()).item()
                assert e1 < 1e-2 and e2 < 1e-2, (key, e1, e2)
            row.append(timeit(call))
            row_nn.append(timeit(call_nn))
            if key[1] >> 8:
                call()
                torch.cuda.synchronize()
                nwg = ((m + 255) // 256) * ((n + 191) // 192)
                buf = (ctypes.c_ulonglong * (8 * nwg))()
                rd = stamp_readers[key]
                assert rd(buf, 8 * nwg) == 0
                import numpy as np
                a = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 8).astype(np.int64)
                t00 = a[:, 0].min()
                print(f'   stamps {name} abl={key[1] & 255}: workgroups {nwg}; start after first start: mean {np.mean(a[:, 0] - t00):.0f} max {np.max(a[:, 0] - t00)}'
                      f' | prologue {np.mean(a[:, 1] - a[:, 0]):.0f} | K loop {np.mean(a[:, 2] - a[:, 1]):.0f} ({a[0, 6]} steps: wait+barrier {np.mean(a[:, 4]):.0f}, issue+multiply {np.mean(a[:, 5]):.0f})'
                      f' | epilogue {np.mean(a[:, 3] - a[:, 2]):.0f} | last end {np.max(a[:, 3]) - t00}  [s_memtime ticks]', flush=True)
        print(f'{name:8s} {"fwd":5s} {t_lib:7.1f} ' + ' '.join(f'{t:7.1f}' for t in row), flush=True)
        print(f'{name:8s} {"dgrad":5s} {t_lib_nn:7.1f} ' + ' '.join(f'{t:7.1f}' for t in row_nn), flush=True)


if __name__ == '__main__':
    build() if sys.argv[1:] == ['build'] else run()
