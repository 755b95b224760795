# validation of HEAD: all GPU tests, smoke(), the rank shapes
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c18
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $OUT/smoke.txt
timeout 600 python tools/rank_shapes.py worlds=1,2,4,8 2>&1 | grep world > $OUT/rank_shapes.txt
cat $OUT/pytest.txt $OUT/smoke.txt $OUT/rank_shapes.txt
