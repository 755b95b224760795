set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c35
mkdir -p $OUT
for c in 1 2; do RLG_DW_CHAIN=$c python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "dw or engine" 2>&1 | tail -2 | tee -a $OUT/tests.log; done
for c in 1 2; do echo "== RLG_DW_CHAIN=$c"; RLG_DW_CHAIN=$c timeout 300 python tools/bench_mlp_chain.py --rows 32768 65536 --no-lib --dw-blocks 1024 --groups 2 2>&1 | grep "dW\|bias col"; done | tee $OUT/bench_chain.log
for c in 1 2 1 2; do RLG_DW_CHAIN=$c python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chain $c', d['ms_per_step'], d['ms_per_step_stats']['min'], d['roofline_mfma']['avg_launch_us'])" | tee -a $OUT/bench.log; done
