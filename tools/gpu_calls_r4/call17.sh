set -x
cd $GRAFT_REPO_ROOT
for B in 0 100 200 400; do RLG_DW_BLOCKS=$B timeout 300 python tools/rank_shapes.py worlds=8 2>&1 | grep world | sed "s/^/RLG_DW_BLOCKS=$B /"; done
for B in 0 100 200; do RLG_DW_BLOCKS=$B timeout 300 python tools/rank_shapes.py worlds=4 2>&1 | grep world | sed "s/^/RLG_DW_BLOCKS=$B /"; done
