# per-phase breakdown of one steady-state step (run on the GPU box: gpurun -- 'bash scratch/phase_cmd.sh')
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_p && mkdir -p gpurun_out/prof_p
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_p -o p -- python3 bench.py --steps 6 --warmup 3 --no-kernel-profile --no-cpu-baseline --no-fp32 > gpurun_out/prof_p/bench.json 2> gpurun_out/prof_p/bench.err
mkdir -p gpurun_out/prof_p/x && mv gpurun_out/prof_p/p_kernel_trace.csv gpurun_out/prof_p/x/ 2>/dev/null
python scratch/trace_phases.py gpurun_out/prof_p 14 > gpurun_out/prof_p/phases.txt 2>&1
rm -rf gpurun_out/prof_p/x
find gpurun_out/prof_p -name '*.db' -delete
cat gpurun_out/prof_p/phases.txt
