set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c3
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -5 | tee $OUT/chain_tests.log
timeout 300 python tools/bench_mlp_chain.py --rows 32768 4096 --no-lib --dw-blocks 1024 2>&1 | tee $OUT/bench_chain.log
