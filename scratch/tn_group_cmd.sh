#!/bin/bash
# gpurun helper: parity of the grouped weight gradients, then the group-vs-per-layer bench at three depths
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PYTHONPATH=.:scratch
timeout 600 python -m pytest tests/test_k17_gemm_gpu.py tests/test_k11_arena_gpu.py -m gpu -x -q > gpurun_out/tn_group_tests.log 2>&1
tail -5 gpurun_out/tn_group_tests.log
for d in 2048 4096 8192; do MBV_GEMM_GROUP_DEPTH=$d timeout 300 python scratch/bench_tn_group.py; done 2>&1 | tee gpurun_out/tn_group_bench.log
