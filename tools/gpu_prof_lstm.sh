cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lstm
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o l -- python $GRAFT_REPO_ROOT/tools/bench_lstm.py > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/l_kernel_trace.csv 28 > $OUT/summary.txt
cat $OUT/summary.txt
rm -f $OUT/l_kernel_trace.csv
