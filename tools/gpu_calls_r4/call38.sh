cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_agent_gpu.py tests/test_mlp_chain_gpu.py -m gpu -q 2>&1 | grep -E "^FAILED|^E  " | cut -c1-260 | head -60
