#!/bin/bash
# round 5, call 4: element-level diff of the two ranks' optimiser arrays behind bench.py --gpus 2 (product library, rowpt launch)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c4; mkdir -p $OUT
export RLG_TEST_SINGLE_GPU=1 RLG_BENCH_PREFLIGHT=0 RLG_ADAM_PACK_ROWPT=1 RLG_BENCH_SYNC_DIFF=1
p=31300
for i in $(seq 1 16); do
  p=$((p+1))
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --steps 1 --warmup 2 > /tmp/o.txt 2> /tmp/e.txt
  echo "run $i in_sync $(grep '^{' /tmp/o.txt | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['config'].get('ranks_in_sync'))")" | tee -a $OUT/diff.txt
  grep -E "^  DIFF|^      " /tmp/e.txt | grep -v identical | cut -c1-500 | tee -a $OUT/diff.txt
done
# the trace library with the trace switched off: does the changed code generation alone remove the desync?
export RLG_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/_build/trace/lib.so
ok=0; bad=0
for i in $(seq 1 12); do
  p=$((p+1))
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --steps 1 --warmup 2 > /tmp/o.txt 2> /tmp/e.txt
  r=$(grep '^{' /tmp/o.txt | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['config'].get('ranks_in_sync'))")
  [ "$r" = "True" ] && ok=$((ok+1)) || bad=$((bad+1))
done
echo "trace library, trace off: in sync $ok, not $bad" | tee -a $OUT/diff.txt
