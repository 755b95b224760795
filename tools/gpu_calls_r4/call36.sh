cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c36; rm -rf $OUT; mkdir -p $OUT
timeout 300 python tools/exp/lean_probe.py 4096 37 2>&1 | grep -E "rows" | cut -c1-300 > $OUT/probe.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/pytest.txt
timeout 600 python tools/rank_shapes.py worlds=1,4,8 2>&1 | grep world > $OUT/rank_shapes.txt
cat $OUT/probe.txt $OUT/pytest.txt $OUT/rank_shapes.txt
