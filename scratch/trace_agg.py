"""Per-step kernel time by name / category from a rocprofv3 kernel trace CSV (last 5 steps, delimited by k_adamw)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step ends with the optimizer pass: one k_adamw launch, or a run of adjacent ones (round 6: the pass is cut around the
# LayerNorm-affine range that K3's backward updates itself) — the LAST launch of a run is the delimiter
idx = [i for i, r in enumerate(rows) if 'k_adamw' in r['Kernel_Name']
       and (i + 1 >= len(rows) or 'k_adamw' not in rows[i + 1]['Kernel_Name'])]
a, b = idx[-6], idx[-1]
seg = rows[a + 1:b + 1]
n = 5


def cat(k):
    if 'Cijk' in k: return 'gemm'
    if 'igemm' in k or 'batched_transpose' in k or 'SubTensor' in k or 'gridwise' in k or 'naive_conv' in k: return 'conv(miopen)'
    if 'at::native' in k:
        for t in ['add<float>', 'add<c10::BFloat16>', 'bfloat16_copy', 'bfloat16tofloat32', 'direct_copy', 'reduce_kernel',
                  'GroupNorm', 'RowwiseMoments', 'Gelu', 'DivFunctor', 'MulFunctor', 'threshold', 'clamp', 'CatArray', 'Fill']:
            if t in k: return 'aten:' + t
        return 'aten:other'
    if 'rocclr' in k: return 'rocclr copy/fill'
    return 'mbv'


agg = collections.defaultdict(lambda: [0, 0]); cats = collections.defaultdict(lambda: [0, 0])
for r in seg:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); k = r['Kernel_Name']
    agg[k[:110]][0] += d; agg[k[:110]][1] += 1
    cats[cat(k)][0] += d; cats[cat(k)][1] += 1
print('launches/step', len(seg) / n, 'busy us/step', sum(v[0] for v in agg.values()) / n / 1e3,
      'wall', (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / n / 1e3)
for k, v in sorted(cats.items(), key=lambda kv: -kv[1][0]): print(f"{v[0]/n/1e3:9.1f} us {v[1]/n:7.1f}  {k}")
print()
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{v[0]/n/1e3:9.1f} us {v[1]/n:7.1f} x {v[0]/v[1]/1e3:8.1f}  {k}")
