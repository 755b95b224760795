#!/bin/bash
# round 5, call 21 (first call of the second session; 20.8 GPU-minutes left): the default bench line, the rocprofv3 kernel-trace
# summary of the same command, then the whole GPU suite on the tree with the round's host-side changes (weights_token, capture
# roll-back, value_size > 1 torch forms, two-rank-vs-oracle test, Runner-on-device tests, derived parity yardstick)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c21; rm -rf $OUT; mkdir -p $OUT
timeout 300 python bench.py 2>$OUT/bench_stderr.txt | tail -1 > $OUT/bench_humanoid.json
python - <<'PY'
import json, os
o = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r5c21/'
try:
    d = json.load(open(o + 'bench_humanoid.json'))
    print('humanoid', round(d['ms_per_step'], 2), 'ms', round(d['value'] / 1e6, 2), 'M; exact', d.get('exact_products_ms_per_step'))
    for k in ('roofline', 'roofline_fwd', 'roofline_fwd_infer', 'roofline_bwd', 'roofline_mfma'):
        r = d[k]; print(' ', k, round(r['avg_launch_us'], 1), 'us frac', round(r['frac'], 3))
    c = d['cpu_baseline']; print('  cpu', c['kind'], round(c['value']), c['cores'], 'threads; ratio', round(d['gpu_over_cpu']))
except Exception as e:
    print('bench line unreadable:', e); print(open(o + 'bench_stderr.txt').read()[-3000:])
PY
( cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact-row --steps 3 --warmup 1 > $OUT/prof_log.txt 2>&1 )
python tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 40 > $OUT/prof_summary.txt 2>&1; cp $OUT/prof/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof
head -12 $OUT/prof_summary.txt
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 400 --tb=short -rf --durations=12 > $OUT/pytest_full.txt 2>&1
echo "pytest rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_full.txt | tail -40
tail -120 $OUT/pytest_full.txt | cut -c1-300 > $OUT/pytest_tail.txt
grep -n "Error\|error\|assert" $OUT/pytest_full.txt | head -60 | cut -c1-300
