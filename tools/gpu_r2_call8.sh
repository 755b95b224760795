set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c8
mkdir -p $OUT
python -m pytest tests/test_headline_gpu.py -m gpu -q --timeout 900 2>&1 | tail -30 | tee $OUT/headline_tests.log
for V in "RLG_EXP_NT=0" "RLG_EXP_NT=1" "RLG_EXP_NT=1 RLG_EXP_GAE_PRETOUCH=1"; do
  echo "== $V"
  env $V timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['ms_per_step'], d['ms_per_step_stats'], d['roofline']['avg_launch_us'], d['roofline']['launch_us_min'], d['roofline']['launch_us_max'], d['roofline']['frac'])
"
done 2>&1 | tee $OUT/gae_exp.log
timeout 600 python bench.py --workload ant --steps 20 --warmup 3 2>&1 | tail -1 | tee $OUT/bench_ant.log
