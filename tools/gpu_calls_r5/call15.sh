#!/bin/bash
# round 5, call 15: truncated planes in the weight-gradient kernels (v_perm instead of v_cvt_pk_bf16_f32): both kernel forms,
# truncated against rounded planes, accuracy + time + phases
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c15; mkdir -p $OUT
R=$GRAFT_REPO_ROOT/tools/exp/_build/dwrne/lib.so
( echo "== register form, truncated planes";  RLG_DW_LDS=0 timeout 300 python tools/exp/dw_bf16_check.py --rows 32768 2>&1 | grep -v "^/opt"
  echo "== register form, rounded planes";    RLG_HIP_LIB=$R RLG_DW_LDS=0 timeout 300 python tools/exp/dw_bf16_check.py --rows 32768 2>&1 | tail -1
  echo "== LDS form, truncated planes";       RLG_DW_LDS=1 timeout 300 python tools/exp/dw_bf16_check.py --rows 32768 2>&1 | grep -v "^/opt"
  echo "== LDS form, rounded planes";         RLG_HIP_LIB=$R RLG_DW_LDS=1 timeout 300 python tools/exp/dw_bf16_check.py --rows 32768 2>&1 | tail -1
  echo "== 16,384 rows: register trunc / LDS trunc"
  RLG_DW_LDS=0 timeout 300 python tools/exp/dw_bf16_check.py --rows 16384 2>&1 | tail -1
  RLG_DW_LDS=1 timeout 300 python tools/exp/dw_bf16_check.py --rows 16384 2>&1 | tail -1
  RLG_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/_build/dwstamps/lib_w0.so timeout 200 python tools/exp/dw_lds_phases.py 2>&1 | grep -v "^/opt"
) 2>&1 | tee $OUT/dw.txt
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x -k "dw" 2>&1 | tail -3 | tee -a $OUT/dw.txt
RLG_DW_LDS=0 timeout 900 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x -k "dw" 2>&1 | tail -3 | tee -a $OUT/dw.txt
