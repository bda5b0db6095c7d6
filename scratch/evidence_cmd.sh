#!/bin/bash
# gpurun helper: the round's evidence in one call — PMC traffic, default bench line, kernel stats, phases, the fp32
# (reference precision) kernel stats + PMC traffic, the other dtype / distribution lines.  Everything lands under
# gpurun_out/ev/ (the GPU test suite: scratch/full_cmd.sh).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; rm -rf gpurun_out/ev; mkdir -p gpurun_out/ev
bash scratch/pmc_cmd.sh > gpurun_out/ev/pmc.log 2>&1; cp gpurun_out/pmc_hbm_traffic.json gpurun_out/pmc_hbm_traffic.csv gpurun_out/ev/
mkdir -p profiles/r06; cp gpurun_out/pmc_hbm_traffic.json profiles/r06/pmc_hbm_traffic.json      # the bench line below reads it
timeout 900 python3 bench.py --detail-out gpurun_out/ev/bench_detail.json > gpurun_out/ev/bench_default.json 2> gpurun_out/ev/bench_default.err; tail -c 300 gpurun_out/ev/bench_default.json
bash scratch/prof_cmd.sh > gpurun_out/ev/prof.log 2>&1
cp gpurun_out/prof_c/seq.txt gpurun_out/ev/kernel_sequence_one_step.txt; cp gpurun_out/prof_c/kernel_stats.csv gpurun_out/ev/kernel_stats.csv; cp gpurun_out/prof_c/agg.txt gpurun_out/ev/kernel_trace_by_step.txt; cp gpurun_out/prof_c/bench.json gpurun_out/ev/bench_under_rocprof.json
bash scratch/phase_cmd.sh > gpurun_out/ev/phase.log 2>&1; cp gpurun_out/prof_p/phases.txt gpurun_out/ev/phases.txt
# fp32: the precision the reference trains in
PROF_OUT=prof_f32 bash scratch/prof_cmd.sh --dtype fp32 > gpurun_out/ev/prof_f32.log 2>&1
cp gpurun_out/prof_f32/kernel_stats.csv gpurun_out/ev/fp32_kernel_stats.csv; cp gpurun_out/prof_f32/agg.txt gpurun_out/ev/fp32_kernel_trace_by_step.txt; cp gpurun_out/prof_f32/bench.json gpurun_out/ev/fp32_bench_under_rocprof.json
PMC_OUT=fp32_pmc_hbm_traffic bash scratch/pmc_cmd.sh --dtype fp32 > gpurun_out/ev/pmc_f32.log 2>&1; cp gpurun_out/fp32_pmc_hbm_traffic.json gpurun_out/fp32_pmc_hbm_traffic.csv gpurun_out/ev/; cp gpurun_out/fp32_pmc_hbm_traffic.json profiles/r06/fp32_pmc_hbm_traffic.json      # (the next bench line's fp32 object reads it)
for cfg in "--dtype fp16" "--distribution uniform"; do
  name=$(echo $cfg | awk '{print $2}')
  timeout 600 python3 bench.py $cfg --no-cpu-baseline --no-kernel-profile --no-fp32 > gpurun_out/ev/bench_$name.json 2> gpurun_out/ev/bench_$name.err
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/ev/bench_$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])"
done
# derived busy counters per kernel (VALUBusy / SALUBusy, MfmaUtil / LDSBankConflict: two more --pmc passes)
bash scratch/pmc_busy_cmd.sh > gpurun_out/ev/pmc_busy.log 2>&1; cp gpurun_out/pmc_b/pmc_busy.csv gpurun_out/ev/pmc_busy.csv
python3 scratch/show_bench.py gpurun_out/ev/bench_default.json 12
head -24 gpurun_out/ev/kernel_trace_by_step.txt
# round 6: RCCL executes the graph step's exchange at world size 1; BASELINE configs[3] / [4] on one GPU
timeout 600 python3 bench.py --force-reducer --steps 60 --warmup 5 --no-cpu-baseline --no-fp32 --no-kernel-profile > gpurun_out/ev/bench_force_reducer.json 2> gpurun_out/ev/bench_force_reducer.err
bash scratch/configs34_cmd.sh > gpurun_out/ev/configs34.log 2>&1; cp gpurun_out/c34/*_bench.json gpurun_out/c34/*_detail.json gpurun_out/c34/*_kernel_trace_by_step.txt gpurun_out/ev/
