#!/bin/bash
# gpurun helper: K20 tests + fp32 model tests, then the same-box A/B of the grouped weight-gradient launch
python3 -m pytest tests/test_k20_gemm32s_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4
python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "fp32 or oracle or loss" 2>&1 | tail -4
bash scratch/ab32_cmd.sh tn32_group=0 tn32_group=1
