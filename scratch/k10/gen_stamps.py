"""Instrumented copy of K10 (scratch): phase stamps from wall_clock64 (100 MHz) accumulated per phase by thread 0 of
every workgroup.  usage (here): python scratch/k10/gen_stamps.py  -> scratch/k10/k10s.so ; on the box: python scratch/k10/run_stamps.py"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, 'mask_bev_amd/csrc/select_points.hip')).read()
src = src.replace('namespace {\n', 'namespace {\n__device__ unsigned long long g_stamps[16];\n'
                  '#define STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long t_now = wall_clock64(); '
                  'atomicAdd(&g_stamps[i], t_now - t_prev); t_prev = t_now; } } while (0)\n', 1)
marks = re.findall(r'// \[phase (\d+)[^\]]*\]', src)
src = re.sub(r'// \[phase (\d+)[^\]]*\]', lambda m: 'STAMP(%s);' % m.group(1), src)
src = src.replace('// [stamps begin]', 'unsigned long long t_prev = wall_clock64();')
src += '''
extern "C" int k10_read_stamps(unsigned long long* out, int reset) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }
  return 0;
}
'''
gen = os.path.join(root, 'scratch/k10/k10s_gen.hip')
open(gen, 'w').write(src)
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-I', os.path.join(root, 'mask_bev_amd/csrc'),
       '-I', os.path.join(root, 'include'), '-o', os.path.join(root, 'scratch/k10/k10s.so'), gen]
print(' '.join(cmd), 'marks', marks)
subprocess.check_call(cmd)
