#!/bin/bash
# round 5, call 13: phase stamps of the LDS-staged dW kernel (workgroup 8: layer (200, 400) tile 0; waves 0 and 3), 2 and 1 workgroups per CU
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c13; mkdir -p $OUT
for w in 0 3; do
  echo "== wave $w, default K-slices (2 workgroups per CU)" | tee -a $OUT/phases.txt
  RLG_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/_build/dwstamps/lib_w$w.so timeout 200 python tools/exp/dw_lds_phases.py 2>&1 | grep -v "^/opt" | tee -a $OUT/phases.txt
  echo "== wave $w, 16 K-slices (1 workgroup per CU)" | tee -a $OUT/phases.txt
  RLG_DW_LDS_KSPLIT=16 RLG_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/_build/dwstamps/lib_w$w.so timeout 200 python tools/exp/dw_lds_phases.py 2>&1 | grep -v "^/opt" | tee -a $OUT/phases.txt
done
