#!/bin/bash
# gpurun helper: the fp32 whole-model parity tests, then the fp32 step with / without K20
mkdir -p gpurun_out/fp32
python3 -m pytest tests/test_model_gpu.py tests/test_k17_gemm_gpu.py tests/test_graph_gpu.py -x -q -p no:cacheprovider -k "not 16bit and not fp16 and not waymo" > gpurun_out/fp32/tests.log 2>&1; echo "fp32 model tests rc=$? $(tail -1 gpurun_out/fp32/tests.log)"; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/fp32/tests.log | head -30
for sw in "gemm32s=1" "gemm32s=0"; do
  timeout 600 python3 bench.py --dtype fp32 --steps 40 --no-cpu-baseline --no-fp32 --switch $sw --detail-out gpurun_out/fp32/detail_$sw.json > gpurun_out/fp32/bench_$sw.json 2> gpurun_out/fp32/bench_$sw.err
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/fp32/bench_$sw.json').read().strip().splitlines()[-1])
print('$sw', round(d['value'],2), 'scans/s', round(d['ms_per_step'],2), 'ms', 'loss', d['config']['final_loss'])
f=json.load(open('gpurun_out/fp32/detail_$sw.json'))
for r in f['roofline_all'][:12]:
    print(f"   {r['kernel']:28s} n={r['launches_per_step']:6.1f} avg={r['avg_ms']*1e3:8.1f}us tot={r['total_ms_per_step']:.3f}ms frac={r['frac']:.3f} {r['bound']}")
PY
done
