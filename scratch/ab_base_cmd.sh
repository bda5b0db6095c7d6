#!/bin/bash
# gpurun helper: same-box A/B of the working tree against the round's baseline checkout under scratch/_base/ (git-ignored;
# made with `git archive <commit> mask_bev_amd bench.py | tar -x -C scratch/_base/` + its own build).  Alternates the two
# ${AB_PAIRS:-2} times; extra arguments go to both bench.py.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
one() { timeout 600 python $1 --steps ${AB_STEPS:-100} --warmup 5 --no-kernel-profile --no-cpu-baseline --no-fp32 "${@:2}" 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3), d['config'].get('final_loss'))"; }
for i in $(seq ${AB_PAIRS:-2}); do
  echo -n "base: "; one scratch/_base/bench.py "$@"
  echo -n "tree: "; one bench.py "$@"
done 2>&1 | tee -a gpurun_out/ab_base.log
