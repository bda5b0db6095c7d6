"""Ordered kernel list of one phase of the last step (rocprofv3 kernel trace)."""
import csv, glob, sys
d, start_marker, end_marker = sys.argv[1], sys.argv[2], sys.argv[3]
thr = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_ln_apply' in r['Kernel_Name']]
step = rows[idx[-2]:idx[-1]]
a = next(i for i, r in enumerate(step) if start_marker in r['Kernel_Name'])
b = next(i for i, r in enumerate(step) if end_marker in r['Kernel_Name'])
t0 = int(step[a]['Start_Timestamp'])
small = 0.0; ns = 0
for r in step[a:b]:
    du = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if du >= thr:
        if ns: print(f'          … {ns} small kernels, {small:.0f} us'); small = 0.0; ns = 0
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.0f} {du:8.1f} us  {r["Kernel_Name"][:130]}')
    else:
        small += du; ns += 1
if ns: print(f'          … {ns} small kernels, {small:.0f} us')
