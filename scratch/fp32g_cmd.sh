#!/bin/bash
mkdir -p gpurun_out/fp32
python3 -m pytest tests/test_k20_gemm32s_gpu.py -x -q -p no:cacheprovider > gpurun_out/fp32/k20.log 2>&1; echo "k20 tests rc=$? $(tail -1 gpurun_out/fp32/k20.log)"; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/fp32/k20.log | head -30
python3 -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_guard_gpu.py -x -q -p no:cacheprovider -k "not 16bit and not fp16 and not waymo and not bf16" > gpurun_out/fp32/tests.log 2>&1; echo "fp32 tests rc=$? $(tail -1 gpurun_out/fp32/tests.log)"; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/fp32/tests.log | head -30
bash scratch/ab32_cmd.sh ffn32=1 ffn32=0
