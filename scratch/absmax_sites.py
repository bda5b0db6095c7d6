"""Which call sites still run an absmax pass in the fp32 step (after the hints), with the bytes they read (eager step)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mask_bev_amd import ops, ops_records, ops_gemm
sites, byts = collections.Counter(), collections.Counter()
orig = ops.f32_absmax
def spy(tensors):
    st = traceback.extract_stack(limit=5)
    key = ('CAPTURE ' if torch.cuda.is_current_stream_capturing() else '') + ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}:{f.name}' for f in reversed(st[:-1]))[:150]
    sites[key] += 1
    byts[key] += sum(t.numel() * 4 for t in tensors)
    return orig(tensors)
# f32_absmax is resolved inside ops_records (operand_amax) and ops_gemm (the K20 wrappers): spy in both
ops_records.f32_absmax = spy
ops_gemm.f32_absmax = spy
orig_group = ops.gemm32s_tn_group
seen = collections.Counter()
def spy_group(items):
    for it in items:
        it = tuple(it) + (None, None) if len(it) == 3 else tuple(it)
        for j in (0, 1):
            if it[3 + j] is None and ops.amax_hint_get(it[j]) is None:
                t = it[j]
                seen[('g' if j == 0 else 'x', tuple(t.shape), t.stride(0), t._base is not None)] += 1
    return orig_group(items)
ops_gemm.gemm32s_tn_group = spy_group
import bench
sys.argv = ['bench.py', '--dtype', 'fp32', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-fp32', '--no-kernel-profile'] + (['--no-graph'] if os.environ.get('EAGER', '1') == '1' else [])
bench.main()
n = 3
print('absmax passes per step by call site (launches, MB):')
for k, v in sorted(byts.items(), key=lambda kv: -kv[1]):
    print(f'  {sites[k] / n:6.1f}  {v / n / 1e6:8.1f} MB  {k}')
print('grouped weight-gradient operands without a record (kind, shape, row stride, is a view) x count over 3 steps:')
for k, v in sorted(seen.items(), key=lambda kv: -kv[0][1][0] * kv[0][1][1] * kv[1])[:20]:
    print('  ', k, v)
