# timing-only ablations (-DRLG_ABL masks, csrc/mlp_chain_common.hpp) of the pipelined 16-row kernels at a rank's sizes
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c28; rm -rf $OUT; mkdir -p $OUT
B=$GRAFT_REPO_ROOT/tools/exp/_build
echo "=== build (no ablation)" >> $OUT/abl.txt
timeout 200 python tools/exp/rank_chain_probe.py 4096 2>/dev/null | grep rows >> $OUT/abl.txt
for m in 64 128 192 4 16 8 196 212; do
  echo "=== RLG_ABL=$m" >> $OUT/abl.txt
  RLG_HIP_LIB=$B/librlg_abl$m.so timeout 200 python tools/exp/rank_chain_probe.py 4096 2>/dev/null | grep rows >> $OUT/abl.txt
done
echo "=== build again" >> $OUT/abl.txt
timeout 200 python tools/exp/rank_chain_probe.py 4096 2>/dev/null | grep rows >> $OUT/abl.txt
cat $OUT/abl.txt
