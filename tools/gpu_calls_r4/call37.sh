cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c37; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $OUT/pytest.txt
timeout 600 python tools/rank_shapes.py worlds=1,2,4,8 2>&1 | grep world > $OUT/rank_shapes.txt
cat $OUT/pytest.txt | cut -c1-220; cat $OUT/rank_shapes.txt
