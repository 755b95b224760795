#!/bin/bash
# round 5, call 19: adam_frags (product build: the flat-range code IS in the kernel) with the flat-range workgroups not launched
# (RLG_FRAGS_NO_TAIL_BLOCKS=1), 12 two-rank runs - code generation or the workgroups' execution?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c19; mkdir -p $OUT
export RLG_TEST_SINGLE_GPU=1
p=32500
for nt in 1 0; do
  ok=0; bad=0
  for i in $(seq 1 12); do
    p=$((p+1))
    RLG_FRAGS_NO_TAIL_BLOCKS=$nt PROBE_NOTRACE=1 PROBE_ENVS=16384 PROBE_MB=8192 RLG_BENCH_CONFIG='{"adam_frags_multi_gpu": true}' timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p tools/exp/adam_trace_probe.py 3 > /tmp/o.txt 2>&1
    if grep -q "all True" /tmp/o.txt; then ok=$((ok+1)); else bad=$((bad+1)); fi
  done
  echo "adam_frags, flat-range workgroups $([ $nt = 1 ] && echo NOT) launched: in sync $ok, not $bad" | tee -a $OUT/tail.txt
done
