#!/bin/bash
# gpurun helper: K19 tests + K5 packed tests, whole-model tests through the fused decoder, the K5 bench, then A/B of the bench step
mkdir -p gpurun_out/k19
python3 -m pytest tests/test_k19_rowchain_gpu.py tests/test_k5_msda_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/k19/tests.log 2>&1; echo "k19+k5 rc=$? $(tail -1 gpurun_out/k19/tests.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/k19/tests.log | head -30; grep -E "^E  " gpurun_out/k19/tests.log | head -40
python3 -m pytest tests/test_model_gpu.py tests/test_k11_arena_gpu.py tests/test_graph_gpu.py tests/test_fp16_gpu.py -q -m gpu -p no:cacheprovider -s > gpurun_out/k19/model.log 2>&1; echo "model rc=$? $(tail -1 gpurun_out/k19/model.log)"; grep -E "^(FAILED|ERROR)" gpurun_out/k19/model.log | head -30; grep -E "^E  " gpurun_out/k19/model.log | head -40; grep -A 16 "whole-model errors" gpurun_out/k19/model.log | head -60
python3 scratch/bench_msda_bwd.py 2.0 2>&1 | tail -7
bash scratch/ab_cmd.sh "MBV_DECODER_FUSED=0 MBV_MSDA_PACKED=0" "MBV_DECODER_FUSED=0 MBV_MSDA_PACKED=1" "MBV_DECODER_FUSED=1 MBV_MSDA_PACKED=1"
