# round-3 record run: the default bench line (with cpu_baseline), the config-#2 and config-#5 lines, the rocprofv3
# kernel-trace summaries of the humanoid and lstm bench commands, per-rank emulation at world 1/2/4/8.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3final
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py 2>&1 | tail -1 | tee $OUT/bench_humanoid.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_humanoid_20.json
timeout 600 python bench.py --workload ant --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_ant.json
timeout 600 python bench.py --workload lstm --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_lstm.json
python tools/rank_shapes.py worlds=1,2,4,8 2>&1 | grep world | tee $OUT/rank_shapes.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 45 > $OUT/prof_summary.txt
cp $OUT/prof/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload lstm --no-cpu-baseline --steps 3 --warmup 1 > $OUT/prof_log_lstm.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 45 > $OUT/prof_summary_lstm.txt
rm -rf $OUT/prof
cat $OUT/prof_summary.txt $OUT/prof_summary_lstm.txt
