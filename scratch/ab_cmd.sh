#!/bin/bash
# gpurun helper: A/B of path selectors (mask_bev_amd/switches.py) on the bench step.
# usage: bash scratch/ab_cmd.sh "name=a name2=b" "name=c" ...   ("" = defaults)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg"
  timeout 400 python bench.py $(for kv in $cfg; do echo --switch $kv; done) --steps ${AB_STEPS:-60} --warmup 5 --no-kernel-profile --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3), d['config'].get('final_loss'))"
done 2>&1 | tee -a gpurun_out/ab.log
