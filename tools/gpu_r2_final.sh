# round-2 record run (after the split-bf16 weight gradients): the default bench line, the config-#2 line, the
# rocprof summary of the bench command, dW accuracy / time of both product forms, per-rank emulation at world 1/2/4/8.
# (The GAE PMC pass - tools/gpu_pmc_bench_gae.sh - and the MLP microbenchmarks of the earlier record run are unchanged.)
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2final
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py 2>&1 | tail -1 | tee $OUT/bench_humanoid.json
timeout 600 python bench.py --workload ant --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_ant.json
for m in 1 0; do RLG_DW_BF16=$m timeout 120 python tools/exp/dw_bf16_check.py --reps 200 2>&1 | grep -v amdgpu.ids | tee -a $OUT/dw_bf16x6.txt; done
python tools/rank_shapes.py worlds=1,2,4,8 2>&1 | grep world | tee $OUT/rank_shapes.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 45 > $OUT/prof_summary.txt
cat $OUT/prof_summary.txt
cp $OUT/prof/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
