# round-4 PMC passes over the split-product launches at 32,768 rows (tools/bench_mlp_chain.py): instruction mix, MFMA-pipe
# busy, LDS conflicts, HBM bytes of mlp_chain_fwd_bx / mlp_chain_bwd_bx / mlp_dw_bf16x6 -> profiles/r4_chain_bx_pmc.txt, r4_dw_pmc.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4pmc
rm -rf $OUT; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 256 --groups 4 --reps 5"
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" \
         "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  D=$OUT/$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o v -- $CMD > /dev/null 2>&1
  rm -f $D/*kernel_trace.csv
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT mlp_chain_fwd_bx mlp_chain_bwd_bx > $OUT/chain_bx.txt
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT mlp_dw > $OUT/dw.txt
rm -rf $OUT/*/
$CMD 2>&1 | tail -12 > $OUT/timings.txt
cat $OUT/chain_bx.txt $OUT/dw.txt $OUT/timings.txt
