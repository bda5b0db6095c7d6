#!/bin/bash
# gpurun helper: the fp32 step with K20, its per-family table and the per-operator view of what is left on the library
mkdir -p gpurun_out/fp32
timeout 600 python3 bench.py --dtype fp32 --steps 40 --no-cpu-baseline --no-fp32 --aten-detail gpurun_out/fp32/aten_detail.json --detail-out gpurun_out/fp32/detail.json > gpurun_out/fp32/bench.json 2> gpurun_out/fp32/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/fp32/bench.json').read().strip().splitlines()[-1])
print('fp32', round(d['value'],2), 'scans/s', round(d['ms_per_step'],2), 'ms', 'loss', d['config']['final_loss'], 'coverage', d['roofline_coverage'])
f=json.load(open('gpurun_out/fp32/detail.json'))
for r in f['roofline_all'][:24]:
    print(f"   {r['kernel']:28s} n={r['launches_per_step']:6.1f} avg={r['avg_ms']*1e3:8.1f}us tot={r['total_ms_per_step']:.3f}ms frac={r['frac']:.3f} {r['bound']}")
rows=json.load(open('gpurun_out/fp32/aten_detail.json'))
mm=[r for r in rows if any(k in r[0] for k in ('mm','matmul','linear','conv'))]
print('library GEMM / conv operators by time:')
for r in mm[:40]:
    print(f"   {r[4]:8.1f}us x{r[3]:5.1f} {r[0]:28s} {r[1][:90]} @ {r[2][-60:]}")
PY
