# PMC passes over the fused MLP chain launches (tools/bench_mlp_chain.py, 32,768 rows): what do the waves wait for?
# One rocprofv3 --pmc pass per counter group (gpurun refuses --pmc together with other trace domains).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_chain
rm -rf $OUT; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 256 --groups ${GROUPS_:-2 4} --reps 5"
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL" \
         "SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
         "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
         "TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  D=$OUT/$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o v -- $CMD > /dev/null 2>&1
  rm -f $D/*kernel_trace.csv
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT mlp_chain > $OUT/summary.txt
rm -rf $OUT/*/
cat $OUT/summary.txt
