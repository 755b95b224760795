set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c15
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -3 | tee $OUT/tests.log
timeout 300 python tools/bench_mlp_chain.py --rows 32768 4096 65536 --no-lib --dw-blocks 1024 2>&1 | tee $OUT/bench_chain.log
timeout 300 python tools/bench_mlp_chain.py --net ant --rows 32768 --no-lib --dw-blocks 1024 2>&1 | tee -a $OUT/bench_chain.log
