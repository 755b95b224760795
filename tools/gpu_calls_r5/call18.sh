#!/bin/bash
# round 5, call 18: adam_frags under multi_gpu at the rank-of-8 shape (fails in 80 - 100 % of the runs): which part of the
# launch matters?  variants 5 (no fragment stores, no flat-range tail), 7 (no fragment stores, matrices as one flat float4 range), 12 runs each
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c18; mkdir -p $OUT
export RLG_TEST_SINGLE_GPU=1
p=32300
for v in 5 7; do
  L=$GRAFT_REPO_ROOT/rl_games_amd/librlg_hip.so; [ $v != 0 ] && L=$GRAFT_REPO_ROOT/tools/exp/_build/frags$v/lib.so
  ok=0; bad=0
  for i in $(seq 1 12); do
    p=$((p+1))
    RLG_HIP_LIB=$L PROBE_NOTRACE=1 PROBE_ENVS=16384 PROBE_MB=8192 RLG_BENCH_CONFIG='{"adam_frags_multi_gpu": true}' timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p tools/exp/adam_trace_probe.py 3 > /tmp/o.txt 2>&1
    if grep -q "all True" /tmp/o.txt; then ok=$((ok+1)); else bad=$((bad+1)); fi
  done
  echo "adam_frags variant $v: in sync $ok, not $bad" | tee -a $OUT/variants.txt
done
