set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c20
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "dw or engine" 2>&1 | tail -3 | tee $OUT/tests.log
timeout 300 python tools/bench_mlp_chain.py --rows 32768 4096 --no-lib --dw-blocks 512 1024 2048 --groups 2 2>&1 | tee $OUT/bench_chain.log
timeout 300 python tools/bench_mlp_chain.py --net ant --rows 32768 --no-lib --dw-blocks 1024 --groups 2 2>&1 | tee -a $OUT/bench_chain.log
python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | tee $OUT/bench.json
