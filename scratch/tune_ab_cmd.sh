#!/bin/bash
# gpurun helper: A/B of TunableOp selection tables on the bench step: bash scratch/tune_ab_cmd.sh TABLE.csv [TABLE2.csv ...]
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for tab in "" "$@"; do
    echo "== table: ${tab:-committed}"
    timeout 400 python3 bench.py ${tab:+--gemm-table $tab} --steps 60 --warmup 5 --no-kernel-profile --no-cpu-baseline --no-fp32 2>&1 | grep '"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3), d['config'].get('tuned_gemm_table'))"
  done
done
