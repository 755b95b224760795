#!/bin/bash
# round 5, call 10: the CURRENT multi_gpu default at a rank-of-8's shape (rlg_adam_step + a fragment pack launch; no adam_frags):
# 40 two-rank runs on one GPU - in sync?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c10; mkdir -p $OUT
export RLG_TEST_SINGLE_GPU=1
p=32100; ok=0; bad=0
for i in $(seq 1 40); do
  p=$((p+1))
  PROBE_NOTRACE=1 PROBE_ENVS=16384 PROBE_MB=8192 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p tools/exp/adam_trace_probe.py 3 > /tmp/o.txt 2>&1
  if grep -q "all True" /tmp/o.txt; then ok=$((ok+1)); else bad=$((bad+1)); grep -E "^RESULT|^  DIFF|^      " /tmp/o.txt | cut -c1-300 | tee -a $OUT/plain.txt; fi
done
grep "^RESULT" /tmp/o.txt | cut -c1-200 | tee -a $OUT/plain.txt
echo "rlg_adam_step + pack launch under multi_gpu, rank-of-8 shape, 2 ranks on one GPU: in sync $ok, not $bad" | tee -a $OUT/plain.txt
