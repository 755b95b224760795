set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c41
mkdir -p $OUT
RLG_TEST_SINGLE_GPU=1 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -5 | tee $OUT/tests.log
for r in 16 32; do RLG_LOSS_ROWS=$r python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -k loss 2>&1 | tail -2 | tee -a $OUT/tests.log; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 12 > $OUT/prof_summary.txt
cat $OUT/prof_summary.txt
tail -1 $OUT/prof_log.txt | cut -c1-300
rm -rf $OUT/prof
