cd $GRAFT_REPO_ROOT
timeout 300 python tools/exp/lean_probe.py 4096 8192 37 2>&1 | tail -32
