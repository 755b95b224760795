cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|error" | tail -8
