set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call16
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; grep -n "passed\|failed" $OUT/pytest.txt | tail -3; grep -n "^FAILED\|^E  " $OUT/pytest.txt | head -20 | cut -c1-250
timeout 600 python tools/rank_shapes.py worlds=1,2,4,8 2>&1 | grep world | tee $OUT/rank_shapes.txt
