#!/bin/bash
python3 -m pytest tests/test_k17_gemm_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
python3 -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do for sw in conv3x3_k17=0 conv3x3_k17=1; do
  timeout 600 python3 bench.py --steps 100 --no-cpu-baseline --no-fp32 --no-kernel-profile --switch $sw 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw', round(d['value'],2), round(d['ms_per_step'],3))"
done; done
