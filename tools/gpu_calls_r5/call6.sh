#!/bin/bash
# round 5, call 6: which neighbourhood of the rowpt launch matters?  product build (variant 0) and variants 1 (vmcnt(0) behind the
# Adam stores), 2 (plane offsets loaded up front), 3 (no plane stores): 40 runs of bench.py --gpus 2 each, desync count
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c6; mkdir -p $OUT
export RLG_TEST_SINGLE_GPU=1 RLG_BENCH_PREFLIGHT=0 RLG_ADAM_PACK_ROWPT=1
p=31500
for v in 0 1 2 3; do
  L=$GRAFT_REPO_ROOT/rl_games_amd/librlg_hip.so; [ $v != 0 ] && L=$GRAFT_REPO_ROOT/tools/exp/_build/rowpt$v/lib.so
  ok=0; bad=0
  for i in $(seq 1 40); do
    p=$((p+1))
    RLG_HIP_LIB=$L timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --steps 1 --warmup 2 > /tmp/o.txt 2> /tmp/e.txt
    r=$(grep '^{' /tmp/o.txt | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['config'].get('ranks_in_sync'))")
    [ "$r" = "True" ] && ok=$((ok+1)) || bad=$((bad+1))
  done
  echo "variant $v: in sync $ok, not $bad" | tee -a $OUT/variants.txt
done
