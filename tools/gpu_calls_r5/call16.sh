#!/bin/bash
# round 5, call 16: HBM traffic and VALU / MFMA shares of the two weight-gradient forms (PMC passes over tools/exp/dw_bf16_check.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c16
rm -rf $OUT; mkdir -p $OUT
for form in 1 0; do
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    D=$OUT/f${form}_$(echo $C | tr ' ' '_' | cut -c1-30)
    RLG_DW_LDS=$form timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o v -- python $GRAFT_REPO_ROOT/tools/exp/dw_bf16_check.py --rows 32768 --reps 5 > /dev/null 2>&1
    rm -f $D/*kernel_trace.csv
  done
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT mlp_dw > $OUT/summary.txt
rm -rf $OUT/*/
cat $OUT/summary.txt
