# round-2 record run: in-situ GAE traffic (PMC), the default bench line, the config-#2 line, rocprof summary,
# MLP microbenchmarks, per-rank emulation at world 1/2/4/8
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2final
rm -rf $OUT; mkdir -p $OUT
bash $GRAFT_REPO_ROOT/tools/gpu_pmc_bench_gae.sh 2>&1 | tail -4 | tee $OUT/pmc_gae.log
cd $GRAFT_REPO_ROOT
cp gpurun_out/gae_pmc_traffic.json profiles/gae_pmc_traffic.json 2>/dev/null
timeout 900 python bench.py 2>&1 | tail -1 | tee $OUT/bench_humanoid.json
timeout 600 python bench.py --workload ant --steps 20 --warmup 3 2>&1 | tail -1 | tee $OUT/bench_ant.json
timeout 300 python tools/bench_mlp_chain.py --rows 32768 4096 65536 --dw-blocks 1024 --groups 4 2 1 2>&1 | grep -v "^/opt" | tee $OUT/mlp_microbench.txt
timeout 300 python tools/bench_mlp_chain.py --net ant --rows 32768 --no-lib --dw-blocks 1024 --groups 4 2 1 2>&1 | grep -v "^/opt" | tee -a $OUT/mlp_microbench.txt
timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 1024 --groups 2 --phases 2>&1 | grep -v "^/opt" | tee $OUT/mlp_phases.txt
python tools/rank_shapes.py worlds=1,2,4,8 2>&1 | grep world | tee $OUT/rank_shapes.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 45 > $OUT/prof_summary.txt
cat $OUT/prof_summary.txt
cp $OUT/prof/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -f $OUT/prof/bench_kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof8 -o r8 -- python $GRAFT_REPO_ROOT/tools/rank_shapes.py worlds=8 > $OUT/prof8_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof8/r8_kernel_trace.csv 30 > $OUT/prof_summary_world8.txt
rm -rf $OUT/prof8 $OUT/prof
