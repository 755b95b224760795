cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c33; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof8 -o r -- python $GRAFT_REPO_ROOT/tools/rank_shapes.py worlds=8 > $OUT/log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof8/r_kernel_trace.csv 14 > $OUT/summary_world8_lean.txt; rm -rf $OUT/prof8
cat $OUT/summary_world8_lean.txt
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest "tests/test_agent_gpu.py::test_hip_graph_replays_are_bit_identical_to_eager_training" -m gpu -q -x 2>&1 | tail -30
