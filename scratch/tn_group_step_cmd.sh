#!/bin/bash
# gpurun helper: the bench step with per-layer / grouped weight gradients
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in "0 4096" "1 4096" "all 2048" "all 4096" "all 1024"; do
  set -- $cfg
  echo "MBV_TN_GROUP=$1 depth=$2"
  MBV_TN_GROUP=$1 MBV_GEMM_GROUP_DEPTH=$2 timeout 400 python bench.py --steps 60 --warmup 5 --no-kernel-profile --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('final_loss'))"
done 2>&1 | tee gpurun_out/tn_group_step.log
