# round 2, call 1: correctness of the fused MLP kernels + microbenchmarks + full suite + bench
set -x
mkdir -p gpurun_out/r2c1
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -25 | tee gpurun_out/r2c1/chain_tests.log
timeout 300 python tools/bench_mlp_chain.py --rows 32768 4096 65536 2>&1 | tee gpurun_out/r2c1/bench_chain.log
timeout 600 python -m pytest tests -m gpu -q --timeout 300 --deselect tests/test_mlp_chain_gpu.py 2>&1 | tail -25 | tee gpurun_out/r2c1/all_tests.log
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/r2c1/bench.log
