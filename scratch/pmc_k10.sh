cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_k10 && mkdir -p gpurun_out/pmc_k10
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  D=gpurun_out/pmc_k10/$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o t -- python3 scratch/run_k10.py > $D.log 2>&1
  F=$(find $D -name '*counter_collection.csv' | head -1)
  python - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if 'k_sample_select' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(f'{k:32s} n={len(v):3d} mean {sum(v)/len(v):.4g}')
except Exception as e:
    print('ERR', e)
PY
done
rm -rf gpurun_out/pmc_k10
