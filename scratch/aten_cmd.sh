#!/bin/bash
mkdir -p gpurun_out/aten
timeout 600 python3 bench.py --steps 60 --no-cpu-baseline --no-fp32 --aten-detail gpurun_out/aten/aten_detail.json --detail-out gpurun_out/aten/detail.json > gpurun_out/aten/bench.json 2> gpurun_out/aten/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/aten/bench.json').read().strip().splitlines()[-1])
print('bf16', round(d['value'],2), 'scans/s', round(d['ms_per_step'],2), 'ms')
rows=json.load(open('gpurun_out/aten/aten_detail.json'))
rows=[r for r in rows if not any(k in r[0] for k in ('mm','matmul','linear','conv'))]
print('non-GEMM ATen operators by time:', sum(r[4] for r in rows), 'us', sum(r[3] for r in rows), 'launches')
for r in rows[:60]:
    print(f"   {r[4]:7.1f}us x{r[3]:5.1f} {r[0]:26s} {r[1][:70]:70s} @ {r[2][-50:]}")
PY
