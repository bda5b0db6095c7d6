"""Phase / kernel breakdown of one steady-state step from a rocprofv3 --kernel-trace CSV."""
import csv, glob, collections, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'k_ln_apply' in n]
a, b = idx[-2], idx[-1]
step = rows[a:b]
def dur(r): return (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
print('step kernels', len(step), 'busy', round(sum(map(dur, step)), 2), 'span', (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6)
def first(s, last=False):
    ii = [i for i, r in enumerate(step) if s in r['Kernel_Name']]
    return (ii[-1] if last else ii[0]) if ii else None
bounds = [0, first('k_window_attn_fwd'), first('k_msda_fwd'), first('k_attn_fwd_split'), first('k_point_sample'), first('k_attn_bwd'),
          first('k_msda_bwd'), first('k_window_attn_bwd'), first('k_ln_bwd_dense'), first('k_adamw') or first('multi_tensor'), len(step)]
labels = ['encoder fwd(+copies)', 'swin fwd', 'pixel decoder fwd', 'tr decoder fwd', 'loss (+start bwd)', 'tr decoder bwd', 'pixdec bwd', 'swin bwd', 'encoder bwd', 'optimizer']
for k, lab in enumerate(labels):
    s = step[bounds[k]:bounds[k + 1]]
    if not s: continue
    print(f'== {lab:24s} n={len(s):5d} busy={sum(map(dur, s)):7.2f} ms span={(int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e6:7.2f}')
    dd = collections.defaultdict(lambda: [0, 0])
    for r in s:
        dd[r['Kernel_Name'][:120]][0] += dur(r); dd[r['Kernel_Name'][:120]][1] += 1
    for kk, (t, n) in sorted(dd.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f'     {t:6.2f} ms n={n:4d} avg {t / n * 1e3:7.1f}us {kk}')
