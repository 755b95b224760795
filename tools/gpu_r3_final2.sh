# round-3 record run (split-bf16 chain): the default bench line (with cpu_baseline), the config-#2 and config-#5 lines,
# the rocprofv3 kernel-trace summary of the humanoid bench command, PMC passes over the split-bf16 forward / backward.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3final2
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py 2>&1 | tail -1 | tee $OUT/bench_humanoid.json
timeout 600 python bench.py --workload ant --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_ant.json
timeout 600 python bench.py --workload lstm --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_lstm.json
RLG_CHAIN_BX=0 timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_humanoid_exact_products.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 45 > $OUT/prof_summary.txt
cp $OUT/prof/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
CMD="python $GRAFT_REPO_ROOT/tools/exp/bx_pmc_driver.py"
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o v -- $CMD > /dev/null 2>&1
  rm -f $D/*kernel_trace.csv
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT bx_kernel pack_planes > $OUT/pmc_summary.txt
rm -rf $OUT/pmc_*/
cat $OUT/prof_summary.txt $OUT/pmc_summary.txt
