set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c9
mkdir -p $OUT
python -m pytest tests/test_headline_gpu.py -m gpu -q --timeout 900 2>&1 | tail -12 | tee $OUT/headline_tests.log
for V in "RLG_EXP_X=0" "RLG_EXP_GAE_PRETOUCH=1"; do
  echo "== $V"
  env $V timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/b.json 2>/dev/null
  python -c "
import json
d = json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['launch_us_min'], d['roofline']['launch_us_max'], d['roofline']['frac'])
"
done 2>&1 | tee $OUT/gae_exp.log
