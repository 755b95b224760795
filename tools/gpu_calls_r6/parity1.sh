cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_headline_gpu.py -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids | tail -80
