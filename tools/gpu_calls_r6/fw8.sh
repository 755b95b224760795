cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py tests/test_headline_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed"
