# round-4 second call: the new 16-row pipelined chain kernels + fused step tail - correctness first, then what they buy
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call2
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt
timeout 300 python tools/exp/fuzz_chain.py 120 21 > $OUT/fuzz.txt 2>&1; tail -3 $OUT/fuzz.txt
timeout 300 python tools/bench_mlp_chain.py --rows 4096 8192 --groups 1 --no-lib --reps 50 > $OUT/chain_micro_pipe.txt 2>&1
RLG_CHAIN_PIPE1=0 timeout 300 python tools/bench_mlp_chain.py --rows 4096 8192 --groups 1 --no-lib --reps 50 > $OUT/chain_micro_unit.txt 2>&1
grep -E "forward|backward" $OUT/chain_micro_pipe.txt $OUT/chain_micro_unit.txt
timeout 600 python tools/rank_shapes.py worlds=1,2,4,8 > $OUT/rank_shapes_new.txt 2>&1; cat $OUT/rank_shapes_new.txt
timeout 600 python tools/rank_shapes.py worlds=1,4,8 fused_step_tail=0 > $OUT/rank_shapes_no_tail.txt 2>&1; cat $OUT/rank_shapes_no_tail.txt
RLG_CHAIN_PIPE1=0 timeout 600 python tools/rank_shapes.py worlds=4,8 > $OUT/rank_shapes_no_pipe1.txt 2>&1; cat $OUT/rank_shapes_no_pipe1.txt
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench.json; cut -c1-400 $OUT/bench.json
timeout 900 python tools/exp/parity_drift.py rank > $OUT/drift_rank.txt 2>&1; cat $OUT/drift_rank.txt
timeout 1200 python tools/exp/parity_drift.py full > $OUT/drift_full.txt 2>&1; cat $OUT/drift_full.txt
