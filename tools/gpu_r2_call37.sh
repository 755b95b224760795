set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c37
mkdir -p $OUT
for r in 16 32; do RLG_LOSS_ROWS=$r python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -k loss 2>&1 | tail -2 | tee -a $OUT/tests.log; done
for r in 64 32 16 64 32 16; do
  RLG_LOSS_ROWS=$r python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('loss rows $r', d['ms_per_step'], d['ms_per_step_stats']['min'])" | tee -a $OUT/bench.log
done
