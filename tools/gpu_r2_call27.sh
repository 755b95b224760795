set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c27
mkdir -p $OUT
RLG_TEST_SINGLE_GPU=1 python -m pytest tests/test_agent_gpu.py -m gpu -q -x --timeout 900 -k "ipc or two_rank" 2>&1 | tail -4 | tee $OUT/tests.log
python tools/rank_shapes.py worlds=1,2,4,8 2>&1 | grep world | tee $OUT/rank_shapes.log
timeout 300 python tools/bench_mlp_chain.py --rows 4096 --no-lib --dw-blocks 64 128 256 512 --groups 1 2 2>&1 | tee $OUT/bench_chain.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r8 -- python $GRAFT_REPO_ROOT/tools/rank_shapes.py worlds=8 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/r8_kernel_trace.csv 30 > $OUT/prof_summary_w8.txt
cat $OUT/prof_summary_w8.txt
rm -f $OUT/prof/r8_kernel_trace.csv
