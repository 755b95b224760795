cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x -k "adam_step_frags" 2>&1 | grep -v amdgpu.ids | tail -3
echo "fused (8 + 1 counters):"; python tools/rank_shapes.py worlds=8 2>&1 | grep -v amdgpu.ids
echo "pair:"; python tools/rank_shapes.py worlds=8 adam_packs_frags=0 2>&1 | grep -v amdgpu.ids
