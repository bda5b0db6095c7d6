#!/bin/bash
# gpurun helper: the K20 PMC passes at two shapes (stage-1 fc1: byte-heavy; stage-3 fc1: matrix-heavy)
for a in "nt 65536 768 192 10" "nt 4096 3072 768 20"; do
  echo "=== $a"
  K20_ARGS="$a" bash scratch/prof_k20_cmd.sh 2>&1 | grep -A40 "^k_gemm32s" | grep -E "k_gemm32s|GRBM_GUI|SQ_BUSY_CY|INSTS_VALU|INSTS_MFMA|MFMA_BUSY|WAIT_ANY|WAIT_INST_ANY|WAVE_CYCLES|ACTIVE_INST_ANY|FETCH|WRITE|SQ_WAVES|LDS_BANK|LDS_IDX|WAIT_INST_LDS"
done
