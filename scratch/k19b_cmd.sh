#!/bin/bash
mkdir -p gpurun_out/k19
python3 -m pytest tests/test_k19_rowchain_gpu.py -q -m gpu -p no:cacheprovider -x > gpurun_out/k19/tests.log 2>&1; echo "k19 rc=$? $(tail -1 gpurun_out/k19/tests.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/k19/tests.log | head -30; grep -E "^E  " gpurun_out/k19/tests.log | head -20
echo "-- timing"; python3 scratch/time_k19.py bf16 2>&1 | tail -6
bash scratch/ab_cmd.sh "MBV_DECODER_FUSED=0" "MBV_DECODER_FUSED=1"
