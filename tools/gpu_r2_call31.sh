set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c31
mkdir -p $OUT
RLG_CHAIN_WAVES=8 python tools/exp/debug_w8.py 2>&1 | grep "BAD" > $OUT/dbg.log; echo "debug_w8 BAD lines: $(wc -l < $OUT/dbg.log)"
RLG_TEST_SINGLE_GPU=1 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -5 | tee $OUT/tests.log
RLG_CHAIN_WAVES=4 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "forward or backward" 2>&1 | tail -2 | tee -a $OUT/tests.log
timeout 300 python tools/bench_mlp_chain.py --rows 32768 4096 --no-lib --dw-blocks 1024 256 --groups 2 1 2>&1 | grep -v "^/opt" | tee $OUT/bench_chain.log
python tools/rank_shapes.py worlds=1,2,4,8 2>&1 | grep world | tee $OUT/rank_shapes.log
