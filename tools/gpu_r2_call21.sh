set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c22
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "dw or engine" 2>&1 | tail -3 | tee $OUT/tests.log
for lib in librlg_hip.so librlg_hip_d4.so; do
  echo "== $lib" | tee -a $OUT/bench_chain.log
  RLG_HIP_LIB=$GRAFT_REPO_ROOT/rl_games_amd/$lib timeout 300 python tools/bench_mlp_chain.py --rows 32768 65536 --no-lib --dw-blocks 512 1024 --groups 2 2>&1 | grep "dW\|bias col" | tee -a $OUT/bench_chain.log
  RLG_HIP_LIB=$GRAFT_REPO_ROOT/rl_games_amd/$lib python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-200 | tee -a $OUT/bench.json
done
