cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x -k "adam_step_frags or adam_step_pack" 2>&1 | grep -v amdgpu.ids | tail -12
timeout 900 python -m pytest tests/test_agent_gpu.py tests/test_headline_gpu.py -m gpu -q -x -k "one_launch_step or hip_graph_replays or rank_8192 or two_rank_training_keeps_ranks_in_sync or folded_launches" 2>&1 | grep -v amdgpu.ids | tail -8
python tools/rank_shapes.py worlds=1,8 2>&1 | grep -v amdgpu.ids
python tools/rank_shapes.py worlds=8 adam_packs_frags=0 2>&1 | grep -v amdgpu.ids
