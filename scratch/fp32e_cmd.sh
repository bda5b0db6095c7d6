#!/bin/bash
mkdir -p gpurun_out/fp32
python3 -m pytest tests/test_k20_gemm32s_gpu.py tests/test_k12_layernorm_gpu.py -x -q -p no:cacheprovider > gpurun_out/fp32/k20.log 2>&1; echo "k20+k12 tests rc=$? $(tail -1 gpurun_out/fp32/k20.log)"; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/fp32/k20.log | head -30
python3 -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_guard_gpu.py -x -q -p no:cacheprovider -k "not 16bit and not fp16 and not waymo and not bf16" > gpurun_out/fp32/tests.log 2>&1; echo "fp32 tests rc=$? $(tail -1 gpurun_out/fp32/tests.log)"; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/fp32/tests.log | head -30
bash scratch/fp32b_cmd.sh 2>&1 | head -34
timeout 600 python3 bench.py --dtype fp32 --steps 40 --no-cpu-baseline --no-fp32 --no-kernel-profile --switch amax_hints=0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('amax_hints=0', round(d['value'],2), round(d['ms_per_step'],2))"
