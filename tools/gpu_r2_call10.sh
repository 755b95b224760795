set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c10
mkdir -p $OUT

for V in "RLG_EXP_GAE_PREWARM=2048" "RLG_EXP_GAE_PREWARM=256" "RLG_EXP_GAE_PREWARM=2048 RLG_EXP_GAE_PRETOUCH=1"; do
  echo "== $V"
  env $V timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/b.json 2>/dev/null
  python -c "
import json
d = json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['launch_us_min'], d['roofline']['launch_us_max'], d['roofline']['frac'])
"
done 2>&1 | grep -v "^import\|^d = \|^print\|^'" | tee $OUT/gae_exp.log
