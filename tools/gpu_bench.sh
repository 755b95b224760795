set -x
python bench.py 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/bench_kernel_trace.csv 45 > $OUT/summary.txt
cat $OUT/summary.txt
cp $OUT/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -f $OUT/bench_kernel_trace.csv
