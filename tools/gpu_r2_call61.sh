OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c61
mkdir -p $OUT
for v in 0 1 2; do python tools/exp/debug_loss_bwd.py 2>&1 | grep "rows" | grep -c "equal False"; done
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q --timeout 600 -k "backward_evaluates" 2>&1 | tail -1
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q --timeout 600 -k "backward_evaluates" 2>&1 | tail -1
python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-330
python tools/rank_shapes.py worlds=8 2>&1 | grep world
