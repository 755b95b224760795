# round-2 record run: in-situ GAE traffic (PMC), the default bench line, the config-#2 line, rocprof summary
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2final
mkdir -p $OUT
bash $GRAFT_REPO_ROOT/tools/gpu_pmc_bench_gae.sh 2>&1 | tail -4 | tee $OUT/pmc_gae.log
cd $GRAFT_REPO_ROOT
cp gpurun_out/gae_pmc_traffic.json profiles/gae_pmc_traffic.json 2>/dev/null
timeout 900 python bench.py 2>&1 | tail -1 | tee $OUT/bench_humanoid.json
timeout 600 python bench.py --workload ant --steps 20 --warmup 3 2>&1 | tail -1 | tee $OUT/bench_ant.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 45 > $OUT/prof_summary.txt
cat $OUT/prof_summary.txt
cp $OUT/prof/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -f $OUT/prof/bench_kernel_trace.csv
