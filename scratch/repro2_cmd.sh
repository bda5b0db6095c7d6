#!/bin/bash
# gpurun helper: hunt for the round-2 SIGABRT — new tests first, then the driver's line in the old (alphabetical) order on
# the cold box, then with the allocator cache off, then with serialized launches + glibc heap checking.
mkdir -p gpurun_out/repro2
python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "xyz_only or full_size or nonsquare_200 or pfn_matches or repeated_destination or applied_twice or epoch_train_loss" > gpurun_out/repro2/new_tests.log 2>&1
echo "new tests rc=$? $(tail -1 gpurun_out/repro2/new_tests.log)"; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/repro2/new_tests.log | head -20
run() { name=$1; shift
  env "$@" python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/repro2/$name.log 2>&1
  rc=$?; echo "$name rc=$rc $(tail -1 gpurun_out/repro2/$name.log)"
  cp gpurun_out/fault.log gpurun_out/repro2/${name}_fault.log 2>/dev/null; cp gpurun_out/last_test.txt gpurun_out/repro2/${name}_last_test.txt 2>/dev/null
  if [ $rc -ne 0 ]; then echo "---- last test:"; cat gpurun_out/last_test.txt; echo "---- tail:"; grep -v '^\[mbv-test\|^\.\[mbv-test' gpurun_out/repro2/$name.log | tail -30; echo "---- fault.log:"; head -30 gpurun_out/fault.log; fi
}
run alpha MBV_TEST_ORDER=alpha
run nocache PYTORCH_NO_CUDA_MEMORY_CACHING=1
run serial AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 MALLOC_CHECK_=3 MALLOC_PERTURB_=165
