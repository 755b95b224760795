cd $GRAFT_REPO_ROOT
export RLG_TEST_SINGLE_GPU=1
B=$GRAFT_REPO_ROOT/tools/exp/_build
p=29700
for v in current bis_oldbx bis_scalar; do
  L=$GRAFT_REPO_ROOT/rl_games_amd/librlg_hip.so; [ $v != current ] && L=$B/$v/lib.so
  ok=0; bad=0
  for i in 1 2 3 4 5 6 7 8; do
    p=$((p+1))
    r=$(RLG_BENCH_PREFLIGHT=0 RLG_HIP_LIB=$L python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --steps 1 --warmup 2 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['config'].get('ranks_in_sync'))")
    [ "$r" = "True" ] && ok=$((ok+1)) || bad=$((bad+1))
  done
  echo "$v: in sync $ok, not $bad"
done
