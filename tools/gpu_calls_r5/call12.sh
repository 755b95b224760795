#!/bin/bash
# round 5, call 12: LDS-staged dW v3 (macro-structured loop, unconditional look-ahead, vec decision outside the loop)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c12; mkdir -p $OUT
( RLG_DW_LDS=1 timeout 300 python tools/exp/dw_bf16_check.py --rows 32768 2>&1 | grep -v "^/opt"
  for k in 16 24 48 64; do echo "ksplit $k"; RLG_DW_LDS_KSPLIT=$k timeout 300 python tools/exp/dw_bf16_check.py --rows 32768 2>&1 | tail -1; done
  RLG_DW_LDS=1 timeout 300 python tools/exp/dw_bf16_check.py --rows 16384 2>&1 | tail -1
) 2>&1 | tee $OUT/dw_lds.txt
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x -k "dw" 2>&1 | tail -3 | tee -a $OUT/dw_lds.txt
