#!/bin/bash
# gpurun helper: K4 tests (split mode), fp32 model tests, same-box A/B of the split mode in the fp32 step
python3 -m pytest tests/test_k4_window_attn_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -12
python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "fp32 or oracle or loss" 2>&1 | tail -4
bash scratch/ab32_cmd.sh k4_split=0 k4_split=1
