#!/bin/bash
# round 5, call 14: LDS-staged dW with the dZ fragment reads one block ahead of the MFMAs: time + phases
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c14; mkdir -p $OUT
( RLG_DW_LDS=1 timeout 300 python tools/exp/dw_bf16_check.py --rows 32768 2>&1 | tail -1
  RLG_DW_LDS_KSPLIT=64 timeout 300 python tools/exp/dw_bf16_check.py --rows 32768 2>&1 | tail -1
  RLG_DW_LDS_KSPLIT=16 timeout 300 python tools/exp/dw_bf16_check.py --rows 32768 2>&1 | tail -1
  RLG_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/_build/dwstamps/lib_w0.so timeout 200 python tools/exp/dw_lds_phases.py 2>&1 | grep -v "^/opt"
  RLG_DW_LDS_KSPLIT=16 RLG_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/_build/dwstamps/lib_w0.so timeout 200 python tools/exp/dw_lds_phases.py 2>&1 | grep -v "^/opt"
) 2>&1 | tee $OUT/dw_lds.txt
