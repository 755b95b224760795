# same-box A/B of the plane split forms (all with the compiler-visible conversion): default (packed residuals, two blocks at a
# time in dW) / ab0 (dW splits block by block, packed within a block) / nopk (scalar residuals everywhere)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c24; rm -rf $OUT; mkdir -p $OUT
B=$GRAFT_REPO_ROOT/tools/exp/_build
for rep in 1 2; do
for v in default ab0 nopk; do
  L=$GRAFT_REPO_ROOT/rl_games_amd/librlg_hip.so; [ $v != default ] && L=$B/$v/librlg_hip_$v.so
  echo "== $v (rep $rep)" >> $OUT/ab.txt
  RLG_HIP_LIB=$L timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 256 --groups 4 --reps 20 2>/dev/null | grep -E "forward|backward|dW|sums" >> $OUT/ab.txt
done; done
cat $OUT/ab.txt
