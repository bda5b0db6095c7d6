#!/bin/bash
# gpurun helper: K20 parity tests, then the micro-benchmark against the library
mkdir -p gpurun_out/k20
python3 -m pytest tests/test_k20_gemm32s_gpu.py -x -q -p no:cacheprovider > gpurun_out/k20/tests.log 2>&1; echo "k20 tests rc=$? $(tail -1 gpurun_out/k20/tests.log)"; grep -E "^(FAILED|ERROR)|^E  " gpurun_out/k20/tests.log | head -30
timeout 600 python3 scratch/bench_gemm32s.py > gpurun_out/k20/bench.log 2>&1; cat gpurun_out/k20/bench.log | tail -30
