#!/bin/bash
mkdir -p gpurun_out/k5
python3 -m pytest tests/test_k5_msda_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/k5/tests.log 2>&1; echo "k5 rc=$? $(tail -1 gpurun_out/k5/tests.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/k5/tests.log | head; grep -E "^E  " gpurun_out/k5/tests.log | head -10
python3 scratch/bench_msda_bwd.py 2.0 2>&1 | tail -7
bash scratch/ab_cmd.sh "MBV_MSDA_PACKED=0" "MBV_MSDA_PACKED=1"
