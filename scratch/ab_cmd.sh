#!/bin/bash
# gpurun helper: A/B of environment switches on the bench step.  usage: bash scratch/ab_cmd.sh "VAR=a VAR2=b" "VAR=c" ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 400 python bench.py --steps ${AB_STEPS:-60} --warmup 5 --no-kernel-profile --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3), d['config'].get('final_loss'))"
done 2>&1 | tee -a gpurun_out/ab.log
