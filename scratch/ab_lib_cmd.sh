cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_k1_k3_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
cp mask_bev_amd/libmaskbev_hip.so /tmp/new.so
for i in 1 2; do
  for w in new old; do
    if [ $w = old ]; then cp scratch/_lib_old.so mask_bev_amd/libmaskbev_hip.so; else cp /tmp/new.so mask_bev_amd/libmaskbev_hip.so; fi
    echo "== $w"
    timeout 400 python3 bench.py --steps 60 --warmup 5 --no-kernel-profile --no-cpu-baseline --no-fp32 2>&1 | grep '"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"
  done
done
cp /tmp/new.so mask_bev_amd/libmaskbev_hip.so
