"""The kernels of ONE step in start order from a rocprofv3 kernel trace CSV (the last full step, delimited by k_adamw):
start (us from the step's first kernel), gap to the previous kernel's end on any stream, duration, stream, short name."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step ends with the optimizer pass: one k_adamw launch, or a run of adjacent ones (round 6: the pass is cut around the
# LayerNorm-affine range that K3's backward updates itself) — the LAST launch of a run is the delimiter
idx = [i for i, r in enumerate(rows) if 'k_adamw' in r['Kernel_Name']
       and (i + 1 >= len(rows) or 'k_adamw' not in rows[i + 1]['Kernel_Name'])]
seg = rows[idx[-2] + 1:idx[-1] + 1]
t0 = int(seg[0]['Start_Timestamp'])
end = t0


def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    m = re.match(r'Cijk_(\w+?)_(\w+?)_.*?(MT\d+x\d+x\d+)', k)
    if m: return 'Cijk %s %s' % (m.group(1) + '_' + m.group(2), m.group(3))
    k = re.sub(r'at::native::', '', k)
    return k[:90]


for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r.get('Queue_Id', r.get('Stream_Id', '?'))
    print(f"{(s - t0) / 1e3:9.1f} gap {(s - end) / 1e3:7.1f} dur {(e - s) / 1e3:7.1f} q{q} {short(r['Kernel_Name'])}")
    end = max(end, e)
