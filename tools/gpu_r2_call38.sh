set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c38
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -4 | tee $OUT/tests.log
RLG_CHAIN_WAVES=4 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "forward" 2>&1 | tail -2 | tee -a $OUT/tests.log
for lib in librlg_hip_prev.so librlg_hip.so; do
  echo "== $lib" | tee -a $OUT/bench_chain.log
  RLG_HIP_LIB=$GRAFT_REPO_ROOT/rl_games_amd/$lib timeout 300 python tools/bench_mlp_chain.py --rows 32768 65536 4096 --no-lib --dw-blocks 1024 --groups 0 2>&1 | grep "forward" | tee -a $OUT/bench_chain.log
done
for lib in librlg_hip_prev.so librlg_hip.so librlg_hip_prev.so librlg_hip.so; do
  RLG_HIP_LIB=$GRAFT_REPO_ROOT/rl_games_amd/$lib python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['ms_per_step_stats']['min'])" | tee -a $OUT/bench.log
done
