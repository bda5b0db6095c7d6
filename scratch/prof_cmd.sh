# kernel-trace + stats profile of the bench step (run on the GPU box: gpurun -- 'bash scratch/prof_cmd.sh')
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
# usage: bash scratch/prof_cmd.sh [extra bench.py arguments]; output directory gpurun_out/${PROF_OUT:-prof_c}
P=gpurun_out/${PROF_OUT:-prof_c}
rm -rf $P && mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P -o c -- python3 bench.py --steps 10 --warmup 3 --no-kernel-profile --no-cpu-baseline --no-fp32 $@ > $P/bench.json 2> $P/bench.err
find $P -type f | head
TRACE=$(find $P -name '*kernel_trace.csv' | head -1)
STATS=$(find $P -name '*kernel_stats.csv' | head -1)
PYTHONPATH=. python scratch/trace_agg.py $TRACE 45 > $P/agg.txt 2>&1
cp $STATS $P/kernel_stats.csv
PYTHONPATH=. python scratch/trace_seq.py $TRACE > $P/seq.txt 2>&1
rm -f $TRACE
find $P -name '*.db' -delete
tail -75 $P/agg.txt
