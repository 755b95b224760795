set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call12
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/exp/rollout_consistency.py > $OUT/rollout_consistency_pipe.txt 2>&1; grep -v "Warn\|amdgpu.ids" $OUT/rollout_consistency_pipe.txt | grep "step\|after" | cut -c1-260
