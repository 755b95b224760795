set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c17
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -3 | tee $OUT/tests.log
timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 256 512 1024 --groups 4 2 --phases 2>&1 | tee $OUT/bench_chain.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/bench.log 2>&1
tail -2 $OUT/bench.log
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py --help > /dev/null 2>&1
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -16 {}' | cut -c1-200
find $OUT/prof -name "*.db" -delete; find $OUT/prof -name "*kernel_trace.csv" -delete
