# round 6 record run: bench lines (default + ant + lstm), rocprofv3 kernel summary of the bench command, the rank-shape emulation
# + its kernel summary, PMC traffic of the in-epoch GAE launch
cd $GRAFT_REPO_ROOT
R=gpurun_out/r6_record_f16; mkdir -p $R
python bench.py > $R/bench.json 2> $R/bench.err; tail -c 300 $R/bench.json; echo
python bench.py --workload ant --no-cpu-baseline > $R/bench_ant.json 2> /dev/null; python bench.py --workload lstm --no-cpu-baseline > $R/bench_lstm.json 2>/dev/null
python tools/rank_shapes.py 2>&1 | grep -v amdgpu.ids > $R/rank_shapes.txt; cat $R/rank_shapes.txt
bash tools/gpu_bench.sh > $R/prof_bench_stdout.txt 2>&1; cp gpurun_out/prof_bench/summary.txt $R/bench_kernel_summary.txt; cp gpurun_out/prof_bench/kernel_stats.csv $R/bench_kernel_stats.csv 2>/dev/null
bash tools/gpu_prof_rank8.sh > $R/world8_kernel_summary.txt 2>&1
bash tools/gpu_pmc_bench_gae.sh > $R/gae_pmc_stdout.txt 2>&1; cp gpurun_out/gae_pmc_traffic.json $R/ 2>/dev/null
tail -5 $R/gae_pmc_stdout.txt
