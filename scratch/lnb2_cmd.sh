#!/bin/bash
python3 -m pytest tests/test_k12_layernorm_gpu.py tests/test_k20_gemm32s_gpu.py tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
bash scratch/ab32_cmd.sh ln_bound_hints=0 ln_bound_hints=1
timeout 600 python3 scratch/absmax_sites.py 2>&1 | grep -A 30 "absmax passes per step"
