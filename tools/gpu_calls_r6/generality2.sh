cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_agent_gpu.py tests/test_discrete_gpu.py tests/test_lstm_gpu.py -m gpu -q -x -k "update_matches_reference_epoch or recurrent_layouts or lstm or discrete" 2>&1 | grep -v amdgpu.ids | tail -15
