#!/bin/bash
# round 5, call 5: more instances of the desync (product library, rowpt launch): which lanes / elements / arrays each time
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c5; mkdir -p $OUT
export RLG_TEST_SINGLE_GPU=1 RLG_BENCH_PREFLIGHT=0 RLG_ADAM_PACK_ROWPT=1 RLG_BENCH_SYNC_DIFF=1
p=31400
for i in $(seq 1 60); do
  p=$((p+1))
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --steps 1 --warmup 2 > /tmp/o.txt 2> /tmp/e.txt
  r=$(grep '^{' /tmp/o.txt | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['config'].get('ranks_in_sync'))")
  if [ "$r" != "True" ]; then echo "run $i in_sync $r" | tee -a $OUT/diff.txt; grep -E "^  DIFF|^      " /tmp/e.txt | grep -v identical | cut -c1-700 | tee -a $OUT/diff.txt; fi
done
echo "done 60 runs" | tee -a $OUT/diff.txt
