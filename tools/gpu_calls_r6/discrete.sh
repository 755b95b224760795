cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_discrete_gpu.py tests/test_agent_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -40
