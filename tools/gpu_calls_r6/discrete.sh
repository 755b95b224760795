cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_discrete_gpu.py tests/test_agent_gpu.py tests/test_runner_gpu.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -60
