set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c28
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py tests/test_ops_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -4 | tee $OUT/tests.log
RLG_CHAIN_WAVES=4 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "forward or backward" 2>&1 | tail -2 | tee -a $OUT/tests.log
RLG_TEST_SINGLE_GPU=1 python -m pytest tests/test_agent_gpu.py tests/test_headline_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -4 | tee -a $OUT/tests.log
for w in 4 8; do echo "== RLG_CHAIN_WAVES=$w"; RLG_CHAIN_WAVES=$w timeout 300 python tools/bench_mlp_chain.py --rows 4096 8192 --no-lib --dw-blocks 256 --groups 1 2>&1 | grep -v "^/opt"; done | tee $OUT/bench_chain.log
python tools/rank_shapes.py worlds=1,4,8 2>&1 | grep world | tee $OUT/rank_shapes.log
