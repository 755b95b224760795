cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_agent_gpu.py -m gpu -q -k "central or recurrent or lstm or checkpoint or multi_agent" 2>&1 | grep -v amdgpu.ids | tail -60
