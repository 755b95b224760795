# conversion as a compiler-visible instruction (hazard fix): determinism probe, all GPU tests, bench kernel timings
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c23; rm -rf $OUT; mkdir -p $OUT
timeout 200 python tools/exp/determinism_probe.py 8 > $OUT/probe.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $OUT/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --no-exact-row 2>/dev/null | tail -1 > $OUT/bench.json
python - <<'PY'
import json, os
o = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r4c23/'
d = json.load(open(o + 'bench.json'))
print('humanoid', round(d['ms_per_step'], 2), 'ms', round(d['value'] / 1e6, 2), 'M')
for k in ('roofline', 'roofline_fwd', 'roofline_fwd_infer', 'roofline_bwd', 'roofline_mfma'):
    r = d[k]; print(' ', k, round(r['avg_launch_us'], 1), 'us frac', round(r['frac'], 3))
PY
grep -E "^run|^lib" $OUT/probe.txt | cut -c1-200; cat $OUT/pytest.txt
