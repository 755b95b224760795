#!/bin/bash
# round 5, call 24: record run of the round's final tree - the whole GPU suite (mlp_chain.hip split, yardstick twins, Dynamo-opaque
# forwards), the default bench line, the rank shapes (worlds 1 and 8), the ant / lstm lines
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c24; rm -rf $OUT; mkdir -p $OUT
timeout 300 python bench.py 2>$OUT/bench_stderr.txt | tail -1 > $OUT/bench_humanoid.json
python - <<'PY'
import json, os
o = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r5c24/'
try:
    d = json.load(open(o + 'bench_humanoid.json'))
    print('humanoid', round(d['ms_per_step'], 2), 'ms', round(d['value'] / 1e6, 2), 'M; exact', d.get('exact_products_ms_per_step'))
    for k in ('roofline', 'roofline_fwd', 'roofline_fwd_infer', 'roofline_bwd', 'roofline_mfma'):
        r = d[k]; print(' ', k, round(r['avg_launch_us'], 1), 'us frac', round(r['frac'], 3))
    c = d['cpu_baseline']; print('  cpu', c['kind'], round(c['value']), c['cores'], 'threads; ratio', round(d['gpu_over_cpu']))
except Exception as e:
    print('bench line unreadable:', e); print(open(o + 'bench_stderr.txt').read()[-3000:])
PY
timeout 200 python tools/rank_shapes.py worlds=1,8 2>&1 | grep world | tee $OUT/rank_shapes.txt
timeout 120 python bench.py --workload ant --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_ant.json
timeout 120 python bench.py --workload lstm --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_lstm.json
python - <<'PY'
import json, os
o = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r5c24/'
for w in ('ant', 'lstm'):
    try:
        e = json.load(open(o + f'bench_{w}.json')); print(w, round(e['ms_per_step'], 2), 'ms', round(e['value'] / 1e6, 2), 'M')
    except Exception as ex:
        print(w, 'unreadable', ex)
PY
timeout 560 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 400 --tb=short -rf --durations=6 > $OUT/pytest_full.txt 2>&1
echo "pytest rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_full.txt | tail -30
grep -n "^E " $OUT/pytest_full.txt | head -40 | cut -c1-400
