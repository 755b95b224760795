#!/bin/bash
# round 6: PMC passes over the split-fp16 chain launches (humanoid network, 32,768 rows; tools/exp/bx_pmc_driver.py) and
# over the weight-gradient launch (tools/exp/dw_bf16_check.py) - separate --pmc runs with --kernel-trace only
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6f_pmc
rm -rf $OUT; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/exp/bx_pmc_driver.py"
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o v -- $CMD > /dev/null 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT bx_kernel > $OUT/pmc_chain.txt
rm -rf $OUT/pmc_*/
CMD2="python $GRAFT_REPO_ROOT/tools/exp/dw_bf16_check.py --reps 5"
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  D=$OUT/pmcdw_$(echo $C | tr ' ' '_' | cut -c1-40)
  RLG_DW_F16=1 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o v -- $CMD2 > /dev/null 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT mlp_dw > $OUT/pmc_dw.txt
rm -rf $OUT/pmcdw_*/
cat $OUT/pmc_chain.txt $OUT/pmc_dw.txt
