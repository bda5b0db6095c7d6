#!/bin/bash
# gpurun helper (round 6): measured lines for BASELINE configs[3] (KITTI 496x432, Q 200) and configs[4]
# (Waymo-scale 1024x1024, Q 300, fp16) on ONE GPU: bench line with the per-family roofline table + kernel trace by step.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/c34
run() {   # name, bench args
  local name=$1; shift
  timeout 900 python3 bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-fp32 --detail-out gpurun_out/c34/${name}_detail.json \
      > gpurun_out/c34/${name}_bench.json 2> gpurun_out/c34/${name}_bench.err
  tail -c 400 gpurun_out/c34/${name}_bench.json; echo
  tail -3 gpurun_out/c34/${name}_bench.err
  PROF_OUT=c34_$name bash scratch/prof_cmd.sh "$@" > gpurun_out/c34/${name}_prof.log 2>&1
  cp gpurun_out/c34_$name/agg.txt gpurun_out/c34/${name}_kernel_trace_by_step.txt
  cp gpurun_out/c34_$name/kernel_stats.csv gpurun_out/c34/${name}_kernel_stats.csv
  head -30 gpurun_out/c34/${name}_kernel_trace_by_step.txt
}
run kitti --workload kitti_496x432 --dtype bf16 --batch 4
run waymo --workload waymo_1024 --dtype fp16 --batch ${WAYMO_BATCH:-4}
