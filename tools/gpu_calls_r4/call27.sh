# same-box A/B: first H request of the split-bf16 backward in front of the loss tile (build) / behind it (hlate)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c27; rm -rf $OUT; mkdir -p $OUT
B=$GRAFT_REPO_ROOT/tools/exp/_build
for rep in 1 2 3; do
for v in default hlate; do
  L=$GRAFT_REPO_ROOT/rl_games_amd/librlg_hip.so; [ $v != default ] && L=$B/$v/librlg_hip_$v.so
  RLG_HIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-exact-row 2>/dev/null | tail -1 > $OUT/b.json
  python - "$v" <<'PY' >> $OUT/ab.txt
import json, os, sys
d = json.load(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r4c27/b.json'))
print(sys.argv[1], 'epoch', round(d['ms_per_step'], 2), 'bwd', round(d['roofline_bwd']['avg_launch_us'], 2), 'fwd', round(d['roofline_fwd']['avg_launch_us'], 2), 'dw', round(d['roofline_mfma']['avg_launch_us'], 2))
PY
done; done
cat $OUT/ab.txt
timeout 600 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x 2>&1 | tail -3
