# packed-residual plane split (v_pk_add_f32): kernel tests + the bench line's kernel timings
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c19
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -5 > $OUT/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --no-exact-row 2>/dev/null | tail -1 > $OUT/bench.json
python - <<'PY'
import json, os
o = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r4c19/'
d = json.load(open(o + 'bench.json'))
print('humanoid', round(d['ms_per_step'], 2), 'ms', round(d['value'] / 1e6, 2), 'M')
for k in ('roofline', 'roofline_fwd', 'roofline_fwd_infer', 'roofline_bwd', 'roofline_mfma'):
    r = d[k]; print(' ', k, round(r['avg_launch_us'], 1), 'us frac', round(r['frac'], 3))
PY
cat $OUT/pytest.txt
