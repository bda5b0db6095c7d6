# kernel trace of scratch/bench_pfn.py (gpurun -- 'bash scratch/prof_pfn_cmd.sh')
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_pfn && mkdir -p gpurun_out/prof_pfn
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pfn -o p -- python3 scratch/bench_pfn.py > gpurun_out/prof_pfn/out.txt 2>&1
STATS=$(find gpurun_out/prof_pfn -name '*kernel_stats.csv' | head -1)
cp $STATS gpurun_out/prof_pfn/kernel_stats.csv
find gpurun_out/prof_pfn -name '*kernel_trace.csv' -delete
find gpurun_out/prof_pfn -name '*.db' -delete
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_pfn/kernel_stats.csv')))
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} n={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f}us tot={float(r['TotalDurationNs'])/1e6:7.2f}ms")
PY
tail -5 gpurun_out/prof_pfn/out.txt
