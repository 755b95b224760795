# lean 16-row kernels as the default of MlpChain: all GPU tests (no -x: the whole list of failures), rank shapes
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c32; rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $OUT/pytest.txt
timeout 600 python tools/rank_shapes.py worlds=1,4,8 2>&1 | grep world > $OUT/rank_shapes.txt
RLG_CHAIN_LEAN=0 timeout 600 python tools/rank_shapes.py worlds=4,8 2>&1 | grep world > $OUT/rank_shapes_nolean.txt
cat $OUT/pytest.txt; echo lean; cat $OUT/rank_shapes.txt; echo no-lean; cat $OUT/rank_shapes_nolean.txt
