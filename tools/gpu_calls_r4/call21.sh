set -x
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c21; rm -rf $OUT; mkdir -p $OUT
timeout 300 python tools/exp/determinism_probe.py 6 > $OUT/pk.txt 2>&1
RLG_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/_build/nopk/librlg_hip_nopk.so timeout 300 python tools/exp/determinism_probe.py 6 > $OUT/nopk.txt 2>&1
RLG_DW_BF16=0 timeout 300 python tools/exp/determinism_probe.py 4 > $OUT/pk_dwf32.txt 2>&1
tail -12 $OUT/pk.txt $OUT/nopk.txt $OUT/pk_dwf32.txt
