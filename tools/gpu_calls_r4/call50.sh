cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -k "lean or one_launch" 2>&1 | tail -4
