#!/bin/bash
# fp32: the graphed step against the eager step, same seeds — the final losses after 12 steps (stale absmax records baked into a graph would show)
for mode in "" "--no-graph"; do
  timeout 600 python3 bench.py --dtype fp32 --steps 12 --warmup 3 --no-cpu-baseline --no-fp32 --no-kernel-profile $mode 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', d['config']['launch'], 'final_loss', d['config']['final_loss'], round(d['value'],1))"
done
