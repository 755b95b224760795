set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call4
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/exp/pipe1_compare.py > $OUT/pipe1_compare.txt 2>&1; grep -v "Warn\|amdgpu.ids" $OUT/pipe1_compare.txt
( for W in 4 8 16; do RLG_PIPE1_WAVES=$W python tools/exp/rank_chain_probe.py 4096 8192; done ) > $OUT/rank_chain_probe.txt 2>&1; grep rows $OUT/rank_chain_probe.txt
for W in 8 16; do RLG_PIPE1_WAVES=$W timeout 300 python tools/rank_shapes.py worlds=4,8 fused_step_tail=0 2>&1 | grep world | sed "s/^/pipe1 waves $W: /"; done | tee $OUT/rank_shapes_waves.txt
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py tests/test_agent_gpu.py tests/test_headline_gpu.py -m gpu -q -k "non_finite or nan_observation or lstm or three_epochs or fused_step_tail" > $OUT/pytest_sel.txt 2>&1; tail -30 $OUT/pytest_sel.txt
