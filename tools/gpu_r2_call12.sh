set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c12
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py tests/test_ops_gpu.py tests/test_headline_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -6 | tee $OUT/tests.log
timeout 300 python tools/bench_mlp_chain.py --rows 32768 4096 --no-lib --groups 2 --dw-blocks 1024 2>&1 | grep -v "forward\|backward" | tee $OUT/bench_dw.log
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee $OUT/bench.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 14 > $OUT/prof_summary.txt
cat $OUT/prof_summary.txt
rm -f $OUT/prof/bench_kernel_trace.csv
