#!/bin/bash
# gpurun helper: K19 tests + K5 packed tests, whole-model tests through the fused decoder, the K5 bench, then A/B of the bench step
mkdir -p gpurun_out/k19
python3 -m pytest tests/test_k19_rowchain_gpu.py tests/test_k5_msda_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/k19/tests.log 2>&1; echo "k19+k5 rc=$? $(tail -1 gpurun_out/k19/tests.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/k19/tests.log | head -30; grep -E "^E  " gpurun_out/k19/tests.log | head -40
python3 -m pytest tests/test_model_gpu.py tests/test_k11_arena_gpu.py tests/test_graph_gpu.py tests/test_fp16_gpu.py tests/test_launcher_gpu.py tests/test_ddp_graph_gpu.py -q -m gpu -p no:cacheprovider -s > gpurun_out/k19/model.log 2>&1; echo "model rc=$? $(tail -1 gpurun_out/k19/model.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/k19/model.log | head -30; grep -E "^E  " gpurun_out/k19/model.log | head -40; grep -A 16 "whole-model errors" gpurun_out/k19/model.log | head -60
python3 scratch/bench_msda_bwd.py 2.0 2>&1 | tail -7
bash scratch/ab_cmd.sh "MBV_DECODER_FUSED=0 MBV_MSDA_PACKED=0" "MBV_DECODER_FUSED=0 MBV_MSDA_PACKED=1" "MBV_DECODER_FUSED=1 MBV_MSDA_PACKED=1"
python3 bench.py --steps 20 --no-cpu-baseline --no-fp32 > gpurun_out/k19/bench_prof.json 2> gpurun_out/k19/bench_prof.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/k19/bench_prof.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for r in d['roofline_all'][:28]:
    print(f"{r['kernel']:28s} n={r['launches_per_step']:6.1f} avg={r['avg_ms']*1e3:8.1f}us tot={r['total_ms_per_step']:.3f}ms frac={r['frac']:.3f} {r['bound']}")
PY
