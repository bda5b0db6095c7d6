"""Instrumented copy of K10 (scratch): phase stamps from wall_clock64 (100 MHz) accumulated per phase by thread 0 of
every workgroup.  usage (here): python scratch/k10/gen_stamps.py  -> scratch/k10/k10s.so ; on the box: python scratch/k10/run_stamps.py"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, 'mask_bev_amd/csrc/select_points.hip')).read()
src = src.replace('namespace {\n', 'namespace {\n__device__ unsigned long long g_stamps[64];\n'
                  '#define STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long t_now = wall_clock64(); '
                  'atomicAdd(&g_stamps[i], t_now - t_prev); t_prev = t_now; } } while (0)\n', 1)
marks = re.findall(r'// \[phase (\S+)[^\]]*\]', src)
src = re.sub(r'// \[phase (\S+)[^\]]*\]', lambda m: 'STAMP(%s);' % m.group(1), src)
src = src.replace('// [stamps begin]', 'unsigned long long t_prev = wall_clock64(); const unsigned long long t_begin = t_prev;')
src = re.sub(r'// \[wavephase (\S+)[^\]]*\]', lambda m: 'if ((threadIdx.x & 63) == 0) atomicAdd(&g_stamps[%s + (threadIdx.x >> 6)], wall_clock64() - t_begin);' % m.group(1), src)
src += '''
extern "C" int k10_read_stamps(unsigned long long* out, int reset) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64);
  if (reset) { unsigned long long z[64] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }
  return 0;
}
'''
variant = sys.argv[1] if len(sys.argv) > 1 else ''
if 'A' in variant:      # conflict-free LDS reads instead of the bilinear gathers
    assert 'v += b.w[c] * tile[b.o[c]];' in src
    src = src.replace('v += b.w[c] * tile[b.o[c]];', 'v += b.w[c] * tile[(c * 64 + tid + b.o[c] * 1024) & 16383];')
if 'B' in variant:      # no generator
    assert 'if constexpr (RNG) xy[u] = uniform_point(stream, p);' in src
    src = src.replace('if constexpr (RNG) xy[u] = uniform_point(stream, p);', 'if constexpr (RNG) xy[u] = make_float2((float)((p * 37) & 1023) * 9.765625e-4f, (float)(p >> 10) * 0.02f + (float)stream * 1e-12f);')
gen = os.path.join(root, 'scratch/k10/k10s_gen%s.hip' % variant)
open(gen, 'w').write(src)
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-I', os.path.join(root, 'mask_bev_amd/csrc'),
       '-I', os.path.join(root, 'include'), '-o', os.path.join(root, 'scratch/k10/k10s%s.so' % variant), gen]
print(' '.join(cmd), 'marks', marks)
subprocess.check_call(cmd)
