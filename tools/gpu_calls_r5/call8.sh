#!/bin/bash
# round 5, call 8: (a) adam_frags under multi_gpu with the optimiser TUs compiled without the SLP vectoriser: 40 two-rank runs
# (32 of 40 desynchronised with the product build, call 7); (b) the LDS-staged weight-gradient kernel: accuracy against fp64 and
# time next to the register form, K-slice sweep, the dW tests
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c8; mkdir -p $OUT
( for m in 1 0; do RLG_DW_LDS=$m timeout 300 python tools/exp/dw_bf16_check.py --rows 32768 2>&1 | grep -v "^/opt"; done
  RLG_DW_LDS=1 timeout 300 python tools/exp/dw_bf16_check.py --rows 16384 2>&1 | tail -2
  RLG_DW_LDS=0 timeout 300 python tools/exp/dw_bf16_check.py --rows 16384 2>&1 | tail -1
  for k in 16 24 40 48 64; do echo "ksplit $k"; RLG_DW_LDS_KSPLIT=$k timeout 300 python tools/exp/dw_bf16_check.py --rows 32768 2>&1 | tail -1; done
) 2>&1 | tee $OUT/dw_lds.txt
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x -k "dw" 2>&1 | tail -5 | tee -a $OUT/dw_lds.txt
export RLG_TEST_SINGLE_GPU=1 RLG_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/_build/noslp/lib.so
p=31900; ok=0; bad=0
for i in $(seq 1 40); do
  p=$((p+1))
  PROBE_NOTRACE=1 PROBE_ENVS=16384 PROBE_MB=8192 RLG_BENCH_CONFIG='{"adam_frags_multi_gpu": true}' timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p tools/exp/adam_trace_probe.py 3 > /tmp/o.txt 2>&1
  if grep -q "all True" /tmp/o.txt; then ok=$((ok+1)); else bad=$((bad+1)); grep -E "^RESULT|^  DIFF" /tmp/o.txt | cut -c1-300 | tee -a $OUT/frags_noslp.txt; fi
done
grep "^RESULT" /tmp/o.txt | cut -c1-200 | tee -a $OUT/frags_noslp.txt
echo "adam_frags under multi_gpu, no-SLP build: in sync $ok, not $bad" | tee -a $OUT/frags_noslp.txt
