#!/bin/bash
# gpurun helper: the bf16 step's ATen tail by operator and call site
mkdir -p gpurun_out/aten16
timeout 600 python3 bench.py --steps 20 --no-cpu-baseline --no-fp32 --aten-detail gpurun_out/aten16/aten_detail.json > gpurun_out/aten16/bench.json 2> gpurun_out/aten16/bench.err
python3 - <<'PY'
import json
rows = json.load(open('gpurun_out/aten16/aten_detail.json'))
el = [r for r in rows if not any(k in r[0] for k in ('mm', 'matmul', 'linear', 'conv'))]
print('non-GEMM aten: us', round(sum(r[4] for r in el), 1), 'launches', sum(r[3] for r in el))
for r in sorted(el, key=lambda r: -r[3] * 1000 - r[4])[:45]:
    print(f"{r[4]:8.1f}us x{r[3]:5.1f} {r[0]:22s} {r[1][:64]:64s} @ {r[2][-48:]}")
PY
