# round 6, last session: record run of the two-waves-per-SIMD chain kernels: GPU suite, bench lines, kernel summary, rank shapes
cd $GRAFT_REPO_ROOT
R=gpurun_out/r6_record_g; mkdir -p $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee $R/suite.txt
python bench.py > $R/bench.json 2> $R/bench.err; tail -c 300 $R/bench.json; echo
python bench.py --workload ant --no-cpu-baseline > $R/bench_ant.json 2> /dev/null; python bench.py --workload lstm --no-cpu-baseline > $R/bench_lstm.json 2>/dev/null
python tools/rank_shapes.py 2>&1 | grep -v amdgpu.ids > $R/rank_shapes.txt; cat $R/rank_shapes.txt
bash tools/gpu_bench.sh > $R/prof_bench_stdout.txt 2>&1; cp gpurun_out/prof_bench/summary.txt $R/bench_kernel_summary.txt; cp gpurun_out/prof_bench/kernel_stats.csv $R/bench_kernel_stats.csv 2>/dev/null
head -8 $R/bench_kernel_summary.txt
