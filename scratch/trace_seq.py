"""Ordered kernel sequence of one steady-state step from a rocprofv3 --kernel-trace CSV: start offset (us), duration,
gap to the previous kernel's end, short name.  python scratch/trace_seq.py <dir> > seq.txt"""
import csv, glob, re, sys
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_ln_apply' in r['Kernel_Name']]
step = rows[idx[-2]:idx[-1]]
t0 = int(step[0]['Start_Timestamp'])
prev_end = t0


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'void ', '', n)
    m = re.match(r'(Cijk_\w+?_MT\d+x\d+x\d+)', n)
    if m:
        return m.group(1)
    n = re.sub(r'at::native::', '', n)
    return n[:110]


for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {(s - prev_end) / 1e3:6.1f}  q{r.get("Queue_Id", "")} {short(r["Kernel_Name"])}')
    prev_end = max(prev_end, e)
