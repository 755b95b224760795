# round-4 record run: the default bench line (reference timed on the box, exact-product row), ant / lstm lines, per-rank
# emulation, rocprofv3 kernel-trace summary of the bench command and of the world-8 rank shape (-> profiles/r4_*)
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4final
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py 2>$OUT/bench_stderr.txt | tail -1 > $OUT/bench_humanoid.json
timeout 600 python bench.py --workload ant --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_ant.json
timeout 600 python bench.py --workload lstm --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_lstm.json
timeout 600 python tools/rank_shapes.py worlds=1,2,4,8 2>&1 | grep world > $OUT/rank_shapes.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact-row --steps 3 --warmup 1 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 40 > $OUT/prof_summary.txt; cp $OUT/prof/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof8 -o r -- python $GRAFT_REPO_ROOT/tools/rank_shapes.py worlds=8 > $OUT/prof_log8.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof8/r_kernel_trace.csv 30 > $OUT/prof_summary_world8.txt; rm -rf $OUT/prof8
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, os
o = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r4final/'
d = json.load(open(o + 'bench_humanoid.json'))
print('humanoid', round(d['ms_per_step'], 2), 'ms', round(d['value'] / 1e6, 2), 'M; exact', d.get('exact_products_ms_per_step'))
for k in ('roofline', 'roofline_fwd', 'roofline_fwd_infer', 'roofline_bwd', 'roofline_mfma'):
    r = d[k]; print(' ', k, round(r['avg_launch_us'], 1), 'us frac', round(r['frac'], 3), 'split ceiling', r.get('split_ceiling_frac'))
c = d['cpu_baseline']; print('  cpu', c['kind'], round(c['value']), c['cores'], 'threads; ratio', round(d['gpu_over_cpu']), 'ref/port', c.get('port_cross_check', {}).get('reference_over_port'))
for w in ('ant', 'lstm'):
    e = json.load(open(o + f'bench_{w}.json')); print(w, round(e['ms_per_step'], 2), 'ms', round(e['value'] / 1e6, 2), 'M')
PY
cat $OUT/rank_shapes.txt; head -14 $OUT/prof_summary.txt; head -9 $OUT/prof_summary_world8.txt
