set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c52
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -k "loss or backward" 2>&1 | tail -6 | tee $OUT/tests.log
python -m pytest tests/test_agent_gpu.py tests/test_headline_gpu.py -m gpu -q --timeout 900 2>&1 | tail -6 | tee -a $OUT/tests.log
for cfg in '{"loss_in_backward": true}' '{"loss_in_backward": false}' '{"loss_in_backward": true}' '{"loss_in_backward": false}'; do
  RLG_BENCH_CONFIG="$cfg" python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['ms_per_step_stats']['min'])" | tee -a $OUT/bench.log
done
