# kernel-trace + stats profile of the bench step (run on the GPU box: gpurun -- 'bash scratch/prof_cmd.sh')
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c && mkdir -p gpurun_out/prof_c
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c -o c -- python3 bench.py --steps 10 --warmup 3 --no-kernel-profile --no-cpu-baseline --no-fp32 > gpurun_out/prof_c/bench.json 2> gpurun_out/prof_c/bench.err
find gpurun_out/prof_c -type f | head
TRACE=$(find gpurun_out/prof_c -name '*kernel_trace.csv' | head -1)
STATS=$(find gpurun_out/prof_c -name '*kernel_stats.csv' | head -1)
PYTHONPATH=. python scratch/trace_agg.py $TRACE 45 > gpurun_out/prof_c/agg.txt 2>&1
cp $STATS gpurun_out/prof_c/kernel_stats.csv
rm -f $TRACE
find gpurun_out/prof_c -name '*.db' -delete
tail -75 gpurun_out/prof_c/agg.txt
