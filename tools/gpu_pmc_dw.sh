# (replaces the round-1 script of the same name - FETCH_SIZE / WRITE_SIZE over tools/bench_dw_mfma.py, profiles/r1_dw_pmc.txt)
# PMC passes over the split-product weight-gradient launch (tools/bench_mlp_chain.py, 32,768 rows, default plan)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_dw
rm -rf $OUT; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 256 --groups 2 --reps 5"
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  D=$OUT/$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o v -- $CMD > /dev/null 2>&1
  rm -f $D/*kernel_trace.csv
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT mlp_dw > $OUT/summary.txt
rm -rf $OUT/*/
cat $OUT/summary.txt
