#!/bin/bash
python3 -m pytest tests/test_k20_gemm32s_gpu.py -x -q -m gpu -p no:cacheprovider -k "conv3x3" 2>&1 | tail -8
timeout 300 python3 scratch/bench_conv3x3.py 2>&1 | grep -v Warn | tail -4
