# full GPU suite + record numbers of the state "pipe1 (8 waves) default, fused tail opt-in"
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call5
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; tail -12 $OUT/pytest.txt
timeout 600 python tools/rank_shapes.py worlds=1,2,4,8 2>&1 | grep world | tee $OUT/rank_shapes.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof8 -o r -- python $GRAFT_REPO_ROOT/tools/rank_shapes.py worlds=8 > $OUT/prof_log8.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof8/r_kernel_trace.csv 30 > $OUT/prof_summary_world8.txt; rm -rf $OUT/prof8; head -12 $OUT/prof_summary_world8.txt
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py 2>$OUT/bench_stderr.txt | tail -1 > $OUT/bench.json; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['value'], d.get('exact_products_ms_per_step'), d['cpu_baseline']['kind'], d['cpu_baseline']['value'], d['cpu_baseline'].get('port_cross_check',{}).get('reference_over_port'), d['config'].get('chain_products'))"
tail -3 $OUT/bench_stderr.txt
