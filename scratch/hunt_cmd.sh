#!/bin/bash
# gpurun helper: new K5 tests + K2 re-check, the K5 backward bench, then the full suite under three poison patterns
mkdir -p gpurun_out/hunt
python3 -m pytest tests/test_k5_msda_gpu.py tests/test_k2_pfn_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/hunt/k5.log 2>&1; echo "k5+k2 rc=$? $(tail -1 gpurun_out/hunt/k5.log)"; grep -E "^E  " gpurun_out/hunt/k5.log | head
python3 scratch/bench_msda_bwd.py 2.0 2>&1 | tee gpurun_out/hunt/msda_bench.txt
python3 scratch/bench_msda_bwd.py 0.3 2>&1 | tail -5 | tee -a gpurun_out/hunt/msda_bench.txt
for pat in ffffffff 7f7f7f7f 80000000; do
  MBV_POISON=$pat python3 -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/hunt/poison_$pat.log 2>&1
  rc=$?; echo "poison $pat rc=$rc $(tail -1 gpurun_out/hunt/poison_$pat.log)"
  cp gpurun_out/fault.log gpurun_out/hunt/poison_${pat}_fault.log 2>/dev/null
  if [ $rc -ne 0 ]; then echo "---- last test:"; cat gpurun_out/last_test.txt; grep -E "^(FAILED|ERROR)" gpurun_out/hunt/poison_$pat.log | head -20; grep -v '^\[mbv-test\|^\.\[mbv-test' gpurun_out/hunt/poison_$pat.log | tail -15; fi
done
