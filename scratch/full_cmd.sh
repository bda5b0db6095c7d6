#!/bin/bash
# gpurun helper: the driver's GPU-suite line, then the default bench line
mkdir -p gpurun_out/full
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/full/gpu_tests.log 2>&1; echo "suite rc=$? $(tail -1 gpurun_out/full/gpu_tests.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/full/gpu_tests.log | head -20; grep -E "^E  " gpurun_out/full/gpu_tests.log | head -30
grep -A 16 "whole-model errors" gpurun_out/full/gpu_tests.log | head -40
timeout 900 python3 bench.py --detail-out gpurun_out/full/bench_detail.json > gpurun_out/full/bench_default.json 2> gpurun_out/full/bench_default.err; python3 - <<'PY'
import json
text = open('gpurun_out/full/bench_default.json').read().strip().splitlines()[-1]
d = json.loads(text)
print('line bytes', len(text), 'bench', d['value'], d['ms_per_step'], 'fp32', {k: d.get('fp32', {}).get(k) for k in ('value', 'ms_per_step')},
      'cpu', {k: d['cpu_baseline'].get(k) for k in ('value', 'cores')})
full = json.load(open('gpurun_out/full/bench_detail.json'))
for r in full['roofline_all'][:14]:
    print(f"{r['kernel']:28s} n={r['launches_per_step']:6.1f} avg={r['avg_ms']*1e3:8.1f}us tot={r['total_ms_per_step']:.3f}ms frac={r['frac']:.3f} {r['bound']}")
PY
