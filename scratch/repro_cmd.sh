#!/bin/bash
# gpurun helper: the driver's exact GPU-suite line (no cd /tmp, no TMPDIR), N times in fresh processes; stops at the
# first non-zero exit and keeps that run's log + fault.log + last_test.txt under gpurun_out/repro/.
N=${1:-3}; shift
mkdir -p gpurun_out/repro
for i in $(seq 1 $N); do
  env "$@" python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/repro/run_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(tail -1 gpurun_out/repro/run_$i.log)"
  cp gpurun_out/fault.log gpurun_out/repro/fault_$i.log 2>/dev/null
  cp gpurun_out/last_test.txt gpurun_out/repro/last_test_$i.txt 2>/dev/null
  if [ $rc -ne 0 ]; then
    echo "---- last test:"; cat gpurun_out/last_test.txt
    echo "---- log tail (non-test lines):"; grep -v '^\[mbv-test' gpurun_out/repro/run_$i.log | tail -40
    echo "---- fault.log:"; head -40 gpurun_out/fault.log
    dmesg 2>/dev/null | tail -5
    break
  fi
done
