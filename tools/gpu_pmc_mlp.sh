# PMC passes over the three MLP launches (tools/bench_mlp_chain.py, 32,768 rows): where do the waves wait?
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mlp
rm -rf $OUT; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 1024 --groups 2 4 --reps 5"
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  D=$OUT/$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o v -- $CMD > /dev/null 2>&1
  rm -f $D/*kernel_trace.csv
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT mlp_dw_kernel mlp_chain > $OUT/summary.txt
rm -rf $OUT/*/
cat $OUT/summary.txt
