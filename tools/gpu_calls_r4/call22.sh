cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c22; rm -rf $OUT; mkdir -p $OUT
B=$GRAFT_REPO_ROOT/tools/exp/_build
timeout 200 python tools/exp/determinism_probe.py 4 > $OUT/pk.txt 2>&1
for v in a0 b0 ab0; do RLG_HIP_LIB=$B/$v/librlg_hip_$v.so timeout 200 python tools/exp/determinism_probe.py 4 > $OUT/$v.txt 2>&1; done
for f in pk a0 b0 ab0; do echo "== $f"; grep -E "^run|^lib" $OUT/$f.txt | cut -c1-200; done
