# ordered kernel sequence + phases of one steady-state step (gpurun -- 'bash scratch/seq_cmd.sh [MBV_TN_GROUP value]')
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export MBV_TN_GROUP=${1:-1}
export MBV_GEMM_GROUP_DEPTH=${2:-4096}
OUT=gpurun_out/prof_seq_$MBV_TN_GROUP
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python3 bench.py --steps 6 --warmup 3 --no-kernel-profile --no-cpu-baseline --no-fp32 > $OUT/bench.json 2> $OUT/bench.err
mkdir -p $OUT/x && mv $OUT/p_kernel_trace.csv $OUT/x/ 2>/dev/null
python scratch/trace_phases.py $OUT 14 > $OUT/phases.txt 2>&1
python scratch/trace_seq.py $OUT > $OUT/seq.txt 2>&1
rm -rf $OUT/x
find $OUT -name '*.db' -delete
grep "^==\|^step" $OUT/phases.txt
