#!/bin/bash
python3 -m pytest tests/test_k4_window_attn_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8
bash scratch/fp32b_cmd.sh 2>&1 | head -30
