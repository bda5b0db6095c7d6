#!/bin/bash
python3 -m pytest tests/test_k6_attention_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8
python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "fp32 or oracle or loss" 2>&1 | tail -3
bash scratch/ab32_cmd.sh k6_split=0 k6_split=1
