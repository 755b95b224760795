set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c18
mkdir -p $OUT
for c in 2 4; do RLG_DW_CHAIN=$c python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "dw or engine" 2>&1 | tail -2 | tee -a $OUT/tests.log; done
timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 1024 2048 --dw-chain 1 2 4 --groups 2 2>&1 | grep -v phases | tee $OUT/bench_chain.log
for c in 1 2 4; do RLG_DW_CHAIN=$c python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chain', $c, d['ms_per_step'], d['ms_per_step_stats']['min'], d['roofline_mfma']['avg_launch_us'])" | tee -a $OUT/bench.log; done
