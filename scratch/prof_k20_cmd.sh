#!/bin/bash
# gpurun helper: PMC passes over one K20 shape (counters in their own runs, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pk20 && mkdir -p gpurun_out/pk20
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TA_[A-Z0-9_]+|GRBM_[A-Z0-9_]+)\b" | sort -u > gpurun_out/pk20/counters.txt; wc -l gpurun_out/pk20/counters.txt
ARGS="scratch/prof_k20.py ${K20_ARGS:-nt 16384 1536 384 20}"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pk20 -o p$i -- python3 $ARGS > gpurun_out/pk20/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pk20/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'gemm32s' not in k and 'absmax' not in k: continue
        k = 'k_gemm32s' if 'gemm32s' in k else 'k_absmax'
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, d in acc.items():
    print(k, 'avg us', sum(dur[k]) / len(dur[k]), 'VGPR', '')
    for c, v in sorted(d.items()):
        print(f'   {c:28s} {sum(v) / len(v):16.1f}')
PY
find gpurun_out/pk20 -name '*.csv' -size +1M -delete
