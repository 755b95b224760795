#!/bin/bash
# round 5, call 7: (a) rowpt variants 4 (no packed-f32 arithmetic: -fno-slp-vectorize) and 5 (row-major items, no plane stores),
# 40 runs of bench.py --gpus 2 each; (b) the lean kernels' Adam + fragments launch (adam_frags_kernel) under multi_gpu, product
# library: 40 two-rank runs at a rank-of-8's shape (8,192 envs, 4,096-row minibatches per rank) + 12 at a rank-of-4's
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c7; mkdir -p $OUT
export RLG_TEST_SINGLE_GPU=1 RLG_BENCH_PREFLIGHT=0
p=31700
for v in 4 5; do
  L=$GRAFT_REPO_ROOT/tools/exp/_build/rowpt$v/lib.so
  ok=0; bad=0
  for i in $(seq 1 40); do
    p=$((p+1))
    RLG_ADAM_PACK_ROWPT=1 RLG_HIP_LIB=$L timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --steps 1 --warmup 2 > /tmp/o.txt 2> /tmp/e.txt
    r=$(grep '^{' /tmp/o.txt | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['config'].get('ranks_in_sync'))")
    [ "$r" = "True" ] && ok=$((ok+1)) || bad=$((bad+1))
  done
  echo "variant $v: in sync $ok, not $bad" | tee -a $OUT/variants.txt
done
frags() {  # envs mb runs
  ok=0; bad=0
  for i in $(seq 1 $3); do
    p=$((p+1))
    PROBE_NOTRACE=1 PROBE_ENVS=$1 PROBE_MB=$2 RLG_BENCH_CONFIG='{"adam_frags_multi_gpu": true}' timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p tools/exp/adam_trace_probe.py 3 > /tmp/o.txt 2>&1
    if grep -q "all True" /tmp/o.txt; then ok=$((ok+1)); else bad=$((bad+1)); grep -E "^RESULT|^  DIFF|^      " /tmp/o.txt | cut -c1-400 | tee -a $OUT/frags.txt; fi
  done
  grep "^RESULT" /tmp/o.txt | cut -c1-200 | tee -a $OUT/frags.txt
  echo "adam_frags under multi_gpu, 2 ranks on one GPU, envs $1 minibatch $2 (totals): in sync $ok, not $bad" | tee -a $OUT/frags.txt
}
frags 16384 8192 40
frags 32768 16384 12
