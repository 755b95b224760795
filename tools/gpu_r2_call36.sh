set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c36
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "folds_loss or engine or dw" 2>&1 | tail -12 | tee $OUT/tests.log
RLG_TEST_SINGLE_GPU=1 python -m pytest tests/test_agent_gpu.py tests/test_headline_gpu.py -m gpu -q --timeout 900 2>&1 | tail -6 | tee -a $OUT/tests.log
for cfg in '{"norm_in_finalize": true}' '{"norm_in_finalize": false}' '{"norm_in_finalize": true}' '{"norm_in_finalize": false}'; do
  RLG_BENCH_CONFIG="$cfg" python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['ms_per_step_stats']['min'])" | tee -a $OUT/bench.log
done
