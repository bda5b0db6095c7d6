#!/bin/bash
# gpurun helper (round 6): the library-GEMM selections of ANOTHER workload's step, added to the committed table.
#   bash scratch/tune_workload_cmd.sh kitti_496x432 bf16      |     ... waymo_1024 fp16
#  1. the step's signatures that the committed table does not hold: two eager steps with TunableOp in look-up mode on the
#     committed table, untuned signatures recorded
#  2. every signature tuned in a process of its own group (10 per process; a group whose process dies is re-run one by one and the
#     signature that kills its process is pinned to Default), operands rotating through 512 MB (cache-cold)
#  3. committed table + new rows -> gpurun_out/tune_<workload>/gemm_merged.csv, and a same-box A/B of the two tables
cd "$GRAFT_REPO_ROOT"; WL=${1:-kitti_496x432}; DT=${2:-bf16}; OUT=gpurun_out/tune_$WL; rm -rf $OUT; mkdir -p $OUT
cp mask_bev_amd/tuned/gemm_gfx950.csv $OUT/base.csv
export PYTORCH_TUNABLEOP_ROCBLAS_ENABLED=0   # hipBLASLt solutions or Default only (tests/test_host_cpu.py)
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_RECORD_UNTUNED=1 \
PYTORCH_TUNABLEOP_UNTUNED_FILENAME=$OUT/untuned.csv PYTORCH_TUNABLEOP_FILENAME=$OUT/base.csv \
  timeout 600 python3 bench.py --workload $WL --dtype $DT --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-profile --no-fp32 > $OUT/record.log 2>&1
U=$(ls $OUT/untuned*.csv 2>/dev/null | head -1); [ -z "$U" ] && { echo "no untuned signatures recorded"; tail -5 $OUT/record.log; exit 1; }
sort -u $U > $OUT/sigs.csv; echo "untuned signatures: $(wc -l < $OUT/sigs.csv)"
split -l 10 -d $OUT/sigs.csv $OUT/grp_
: > $OUT/results.csv
for g in $OUT/grp_*; do
  if timeout 180 python3 scratch/tune_one.py $g $g.out.csv 512 > $g.log 2>&1; then
    grep -v '^Validator' $g.out*.csv >> $OUT/results.csv
  else
    echo "group $g failed: one by one"
    while read -r line; do
      echo "$line" > $OUT/one.csv; rm -f $OUT/one.out*.csv
      if timeout 60 python3 scratch/tune_one.py $OUT/one.csv $OUT/one.out.csv 512 > $OUT/one.log 2>&1; then
        grep -v '^Validator' $OUT/one.out*.csv >> $OUT/results.csv
      else
        echo "PINNED: $line"; echo "$line" | awk -F, '{print $1","$2",Default,0"}' >> $OUT/results.csv
      fi
    done < $g
  fi
done
cp mask_bev_amd/tuned/gemm_gfx950.csv $OUT/gemm_merged.csv; sort -u $OUT/results.csv >> $OUT/gemm_merged.csv
echo "rows: committed $(wc -l < mask_bev_amd/tuned/gemm_gfx950.csv) merged $(wc -l < $OUT/gemm_merged.csv); Default among the new: $(grep -c Default $OUT/results.csv)"
for t in mask_bev_amd/tuned/gemm_gfx950.csv $OUT/gemm_merged.csv mask_bev_amd/tuned/gemm_gfx950.csv $OUT/gemm_merged.csv; do
  echo -n "$t: "; timeout 600 python3 bench.py --workload $WL --dtype $DT --gemm-table $t --steps 30 --warmup 5 --no-kernel-profile --no-cpu-baseline --no-fp32 2>&1 | grep '"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"
done
