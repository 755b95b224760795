set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c24
mkdir -p $OUT
for cfg in '{"fused_loss": true}' '{"fused_loss": false}' '{"fused_loss": true}' '{"fused_loss": false}'; do
  RLG_BENCH_CONFIG="$cfg" python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['ms_per_step_stats']['min'])" | tee -a $OUT/bench.log
done
