# Adam + plane pack with one thread per (row, 4 columns): kernel / agent tests, bench line, kernel trace of the optimiser launches
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c20
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py tests/test_ops_gpu.py tests/test_agent_gpu.py -m gpu -q -x 2>&1 | tail -5 > $OUT/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --no-exact-row 2>/dev/null | tail -1 > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact-row --steps 3 --warmup 1 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 12 > $OUT/prof_summary.txt; rm -rf $OUT/prof
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, os
o = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r4c20/'
d = json.load(open(o + 'bench.json'))
print('humanoid', round(d['ms_per_step'], 2), 'ms', round(d['value'] / 1e6, 2), 'M')
for k in ('roofline', 'roofline_fwd', 'roofline_fwd_infer', 'roofline_bwd', 'roofline_mfma'):
    r = d[k]; print(' ', k, round(r['avg_launch_us'], 1), 'us frac', round(r['frac'], 3))
PY
cat $OUT/pytest.txt $OUT/prof_summary.txt
