# round 2, call 2: k-outer fused MLP kernels - correctness, microbench, PMC counters
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c2
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -15 | tee $OUT/chain_tests.log
timeout 300 python tools/bench_mlp_chain.py --rows 32768 4096 65536 2>&1 | tee $OUT/bench_chain.log
cd /tmp && export TMPDIR=/tmp
i=0
for GROUP in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/pmc$i -o v -- python $GRAFT_REPO_ROOT/tools/bench_mlp_chain.py --rows 32768 --reps 3 --no-lib --groups 4 2 --dw-blocks 512 > /dev/null 2>&1
  rm -f $OUT/pmc$i/*/v_kernel_trace.csv $OUT/pmc$i/v_kernel_trace.csv
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT mlp_chain mlp_dw_kernel 2>&1 | tee $OUT/pmc_summary.txt
find $OUT -name "*counter_collection.csv" -delete
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600 | tee $OUT/bench.log
