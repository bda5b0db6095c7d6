# derived busy counters per kernel of the bench step (run on the GPU box: gpurun -- 'bash scratch/pmc_busy_cmd.sh')
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_b && mkdir -p gpurun_out/pmc_b
ARGS="bench.py --steps 4 --warmup 2 --no-kernel-profile --no-cpu-baseline --no-fp32"
rocprofv3 --kernel-trace --pmc VALUBusy SALUBusy --output-format csv -d gpurun_out/pmc_b -o a -- python3 $ARGS > gpurun_out/pmc_b/a.json 2> gpurun_out/pmc_b/a.err
rocprofv3 --kernel-trace --pmc MfmaUtil LDSBankConflict --output-format csv -d gpurun_out/pmc_b -o b -- python3 $ARGS > gpurun_out/pmc_b/b.json 2> gpurun_out/pmc_b/b.err
python3 - <<'PY'
import csv, collections, glob, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_b/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        n = n.replace('void ', '').replace('(anonymous namespace)::', '')
        n = re.sub(r'[(<].*', '', n)
        m = re.match(r'_ZN12_GLOBAL__N_1\d+(k_[a-z0-9_]+)', n)
        if m: n = m.group(1)
        if n.startswith('Cijk'): n = 'hipBLASLt (Cijk_*)'
        if n.startswith('at::native') or 'elementwise' in n or 'reduce_kernel' in n: n = 'ATen'
        out[n[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for d in out.values() for c in d})
with open('gpurun_out/pmc_b/pmc_busy.csv', 'w') as fo:
    fo.write('kernel,launches,' + ','.join(names) + '\n')
    for k, d in sorted(out.items(), key=lambda kv: -max(len(v) for v in kv[1].values())):
        n = max(len(v) for v in d.values())
        fo.write(k + ',' + str(n) + ',' + ','.join(f'{sum(d[c]) / len(d[c]):.1f}' if d.get(c) else '' for c in names) + '\n')
print(open('gpurun_out/pmc_b/pmc_busy.csv').read()[:6000])
PY
find gpurun_out/pmc_b -name '*.csv' -size +1M -delete
