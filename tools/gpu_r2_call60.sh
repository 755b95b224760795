OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c60
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q --timeout 600 2>&1 | tail -2
RLG_CHAIN_WAVES=4 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q --timeout 600 -k "forward or backward" 2>&1 | tail -1
python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-330
python tools/rank_shapes.py worlds=8 2>&1 | grep world
