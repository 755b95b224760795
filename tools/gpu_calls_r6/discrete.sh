cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_agent_gpu.py -m gpu -q -k "update_matches_reference or value_size or sigma" 2>&1 | grep -v amdgpu.ids | tail -60
