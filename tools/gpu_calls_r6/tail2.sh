cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/tools/exp/_build
RLG_HIP_LIB=$B/tail_fences.so timeout 600 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x -k "adam_step_frags" 2>&1 | grep -v amdgpu.ids | tail -3
echo "fences:"; RLG_HIP_LIB=$B/tail_fences.so python tools/rank_shapes.py worlds=8 2>&1 | grep -v amdgpu.ids
echo "atomics:"; python tools/rank_shapes.py worlds=8 2>&1 | grep -v amdgpu.ids
echo "pair:"; python tools/rank_shapes.py worlds=8 adam_packs_frags=0 2>&1 | grep -v amdgpu.ids
