# HBM traffic per kernel: two separate PMC passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes (run on the GPU box: gpurun -- 'bash scratch/pmc_cmd.sh')
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w && mkdir -p gpurun_out/pmc_f gpurun_out/pmc_w
# usage: bash scratch/pmc_cmd.sh [extra bench.py arguments, e.g. --dtype fp32] — output gpurun_out/${PMC_OUT:-pmc_hbm_traffic}.json / .csv
OUT=${PMC_OUT:-pmc_hbm_traffic}
ARGS="bench.py --steps 4 --warmup 2 --no-kernel-profile --no-cpu-baseline --no-fp32 $@"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 $ARGS > gpurun_out/pmc_f/bench.json 2> gpurun_out/pmc_f/bench.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 $ARGS > gpurun_out/pmc_w/bench.json 2> gpurun_out/pmc_w/bench.err
F=$(find gpurun_out/pmc_f -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/pmc_w -name '*counter_collection.csv' | head -1)
python scratch/pmc_to_json.py $F $W gpurun_out/$OUT.json gpurun_out/$OUT.csv
find gpurun_out/pmc_f gpurun_out/pmc_w -name '*.csv' -size +2M -delete
find gpurun_out/pmc_f gpurun_out/pmc_w -name '*.db' -delete
cat gpurun_out/$OUT.csv
