#!/bin/bash
# gpurun helper: re-select the library GEMM solutions of the bf16 bench step, cache-cold.
#  1. the step's signatures: two eager steps with TunableOp in look-up mode on an EMPTY table, untuned signatures recorded
#  2. every signature tuned in a process of its own group (10 per process; a group whose process dies is re-run one by one
#     and the signature that kills its process is pinned to Default), operands rotating through 512 MB
#  3. merged table -> gpurun_out/tune/gemm_cold.csv (validator lines from the committed table)
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/tune; rm -rf $OUT; mkdir -p $OUT
DT=${1:-bf16}
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_RECORD_UNTUNED=1 \
PYTORCH_TUNABLEOP_UNTUNED_FILENAME=$OUT/untuned.csv PYTORCH_TUNABLEOP_FILENAME=$OUT/empty.csv \
  timeout 600 python3 bench.py --dtype $DT --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-profile --no-fp32 > $OUT/record.log 2>&1
ls $OUT; U=$(ls $OUT/untuned*.csv | head -1); sort -u $U > $OUT/sigs.csv; wc -l $OUT/sigs.csv
split -l 10 -d $OUT/sigs.csv $OUT/grp_
: > $OUT/results.csv
for g in $OUT/grp_*; do
  if timeout 120 python3 scratch/tune_one.py $g $g.out.csv 512 > $g.log 2>&1; then
    grep -v '^Validator' $g.out*.csv >> $OUT/results.csv
  else
    echo "group $g failed: one by one"
    while read -r line; do
      echo "$line" > $OUT/one.csv; rm -f $OUT/one.out*.csv      # (a process that dies has still appended what it finished)
      if timeout 60 python3 scratch/tune_one.py $OUT/one.csv $OUT/one.out.csv 512 > $OUT/one.log 2>&1; then
        grep -v '^Validator' $OUT/one.out*.csv >> $OUT/results.csv
      else
        echo "PINNED: $line"; echo "$line" | awk -F, '{print $1","$2",Default,0"}' >> $OUT/results.csv
      fi
    done < $g
  fi
done
grep '^Validator' mask_bev_amd/tuned/gemm_gfx950.csv > $OUT/gemm_cold.csv; sort -u $OUT/results.csv >> $OUT/gemm_cold.csv
wc -l $OUT/gemm_cold.csv; grep -c Default $OUT/gemm_cold.csv
