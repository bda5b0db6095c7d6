#!/bin/bash
# gpurun helper: A/B of path selectors on the fp32 step.  usage: bash scratch/ab32_cmd.sh "" "name=value" ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== fp32 $cfg"
  timeout 600 python bench.py --dtype fp32 $(for kv in $cfg; do echo --switch $kv; done) --steps ${AB_STEPS:-30} --warmup 4 --no-kernel-profile --no-cpu-baseline --no-fp32 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3), d['config'].get('final_loss'))"
done 2>&1 | tee -a gpurun_out/ab.log
