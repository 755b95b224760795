cd $GRAFT_REPO_ROOT
timeout 300 python tools/exp/lean_probe.py 4096 --phases 2>&1 | tail -22
