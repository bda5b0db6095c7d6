#!/bin/bash
# gpurun helper: same-box A/B of fp32 switches: bash scratch/ab32_cmd.sh "name=a" "name=b" ...  (each twice, interleaved)
for rep in 1 2; do
for sw in "$@"; do
  timeout 600 python3 bench.py --dtype fp32 --steps 40 --no-cpu-baseline --no-fp32 --no-kernel-profile --switch $sw 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw', round(d['value'],2), round(d['ms_per_step'],2))"
done
done
