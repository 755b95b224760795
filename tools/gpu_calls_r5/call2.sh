#!/bin/bash
# round 5, call 2: the row-per-thread Adam + planes launch in the PRODUCT library (no instrumentation): does it still desync?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c2; mkdir -p $OUT
export RLG_TEST_SINGLE_GPU=1 RLG_BENCH_PREFLIGHT=0
p=31100
for mode in 1 0; do
  ok=0; bad=0
  for i in $(seq 1 8); do
    p=$((p+1))
    RLG_ADAM_PACK_ROWPT=$mode timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --steps 1 --warmup 2 > /tmp/o.txt 2> /tmp/e.txt
    r=$(grep '^{' /tmp/o.txt | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['config'].get('ranks_in_sync'))")
    if [ "$r" = "True" ]; then ok=$((ok+1)); else bad=$((bad+1)); grep "parameter probe" /tmp/e.txt | cut -c1-330; fi
  done
  echo "bench --gpus 2, RLG_ADAM_PACK_ROWPT=$mode: in sync $ok, not $bad" | tee -a $OUT/sync.txt
done
for i in 1 2 3 4 5 6; do
  p=$((p+1))
  RLG_ADAM_PACK_ROWPT=1 PROBE_SYNC=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p tools/exp/two_rank_planes_probe.py 3 2>&1 | grep -E "^epoch|^   " | tail -6 | cut -c1-600 | tee -a $OUT/sync.txt
done
