"""print the interesting parts of a bench.py JSON line: python scratch/show_bench.py FILE [rows]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
if 'roofline_all' not in d:      # round 5 on: the printed line is the summary; the tables sit beside it
    import os
    detail = sys.argv[1].replace('bench_default.json', 'bench_detail.json')
    if os.path.exists(detail):
        d = json.load(open(detail))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
print('bench', round(d['value'], 2), 'scans/s', round(d['ms_per_step'], 3), 'ms; coverage', d.get('roofline_coverage'))
if d.get('fp32'):
    f = d['fp32']
    print('fp32', {k: f.get(k) for k in ('value', 'ms_per_step', 'roofline_coverage')}, (f.get('roofline') or {}).get('kernel'))
    for r in (f.get('roofline_all') or [])[:8]:
        print('   ', r['kernel'], round(r['total_ms_per_step'], 3), 'ms frac', round(r['frac'], 3), r['bound'])
if d.get('cpu_baseline'):
    print('cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('sample', '')[:150])
for r in (d.get('roofline_all') or [])[:n]:
    print(f"{r['kernel']:28s} n={r['launches_per_step']:7.1f} avg={r['avg_ms']*1e3:8.1f}us tot={r['total_ms_per_step']:.3f}ms "
          f"frac={r['frac']:.3f} {r['bound']} traffic={r.get('traffic')}")
