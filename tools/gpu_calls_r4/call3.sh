set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call3
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python tools/exp/tail_check.py > $OUT/tail_check.txt 2>&1; grep -v Warn $OUT/tail_check.txt
RLG_CHAIN_PIPE1=0 timeout 600 python tools/exp/tail_check.py > $OUT/tail_check_nopipe1.txt 2>&1; grep -v Warn $OUT/tail_check_nopipe1.txt | tail -28
( RLG_CHAIN_PIPE1=0 python tools/exp/rank_chain_probe.py 4096 8192
  RLG_CHAIN_PIPE1=0 RLG_CHAIN_WAVES=4 python tools/exp/rank_chain_probe.py 4096
  python tools/exp/rank_chain_probe.py 4096 8192 --phases
  RLG_PIPE1_WAVES=8 python tools/exp/rank_chain_probe.py 4096 8192 --phases ) > $OUT/rank_chain_probe.txt 2>&1
grep -v Warn $OUT/rank_chain_probe.txt
cd /tmp && export TMPDIR=/tmp
for V in 4 8; do
  RLG_PIPE1_WAVES=$V rocprofv3 --kernel-trace --output-format csv -d $OUT/prof$V -o r -- python $GRAFT_REPO_ROOT/tools/rank_shapes.py worlds=8 fused_step_tail=0 > $OUT/prof_log$V.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof$V/r_kernel_trace.csv 12 > $OUT/prof_summary_world8_pipe1_w$V.txt
  rm -rf $OUT/prof$V; head -9 $OUT/prof_summary_world8_pipe1_w$V.txt
done
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_headline_gpu.py::test_three_epochs_on_config2_stay_on_the_oracle_trajectory > $OUT/pytest.txt 2>&1; tail -8 $OUT/pytest.txt
