set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call15
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "one_launch_step or ppo_loss_like or rccl_wrapper or loss" > $OUT/pytest_new.txt 2>&1; tail -25 $OUT/pytest_new.txt | cut -c1-250
timeout 600 python tools/rank_shapes.py worlds=4,8 2>&1 | grep world | tee $OUT/rank_shapes.txt
timeout 600 python tools/rank_shapes.py worlds=4,8 fused_step16=0 2>&1 | grep world | tee -a $OUT/rank_shapes.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof8 -o r -- python $GRAFT_REPO_ROOT/tools/rank_shapes.py worlds=8 > $OUT/prof_log8.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof8/r_kernel_trace.csv 10 > $OUT/prof_summary_world8.txt; rm -rf $OUT/prof8; head -12 $OUT/prof_summary_world8.txt
