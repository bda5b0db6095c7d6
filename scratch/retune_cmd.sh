# WARNING: on ROCm 7.0 one hipBLASLt candidate faults while being measured (GPU memory access fault, the process aborts
# after the signatures tuned so far were saved); the extended table it produced gave no measurable gain over the
# committed one (130.0 / 131.3 vs 132.2 / 131.1 scans/s on one box), which is why the committed table was kept.
# extend the hipBLASLt solution table with the GEMM signatures of the current step (TunableOp, tuning mode), starting
# from the committed table so that known entries (and the one pinned to Default) are kept
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tune
cp mask_bev_amd/tuned/gemm_gfx950.csv gpurun_out/tune/gemm0.csv
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_ROCBLAS_ENABLED=0 \
PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=12 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=2 \
PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tune/gemm.csv timeout 900 python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-profile > gpurun_out/tune/tune.log 2>&1
echo "tuning rc $?"
wc -l gpurun_out/tune/*.csv
