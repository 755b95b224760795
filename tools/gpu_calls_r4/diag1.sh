# round-4 first call: where the 4,096- / 8,192-row (rank-shape) step spends its time, on this round's starting code
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4diag1
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_mlp_chain.py --rows 4096 8192 --groups 1 --phases --no-lib --reps 50 > $OUT/chain_phases_w8.txt 2>&1
RLG_CHAIN_WAVES=4 timeout 300 python tools/bench_mlp_chain.py --rows 4096 --groups 1 --phases --no-lib --reps 50 > $OUT/chain_phases_w4.txt 2>&1
timeout 600 python tools/rank_shapes.py worlds=1,2,4,8 > $OUT/rank_shapes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for W in 8 4; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/prof$W -o r -- python $GRAFT_REPO_ROOT/tools/rank_shapes.py worlds=$W > $OUT/prof_log$W.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof$W/r_kernel_trace.csv 30 > $OUT/prof_summary_world$W.txt
  rm -rf $OUT/prof$W
done
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench.json
tail -5 $OUT/rank_shapes.txt; cat $OUT/bench.json | cut -c1-600
