cd $GRAFT_REPO_ROOT
export RLG_TEST_SINGLE_GPU=1 RLG_BENCH_PREFLIGHT=0
p=29800
run() {  # $1 label, $2 config json
  ok=0; bad=0
  for i in 1 2 3 4 5 6 7 8; do
    p=$((p+1))
    RLG_BENCH_CONFIG="$2" python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --steps 1 --warmup 2 > /tmp/o.txt 2> /tmp/e.txt
    r=$(grep '^{' /tmp/o.txt | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['config'].get('ranks_in_sync'))")
    if [ "$r" = "True" ]; then ok=$((ok+1)); else bad=$((bad+1)); grep "parameter probe" /tmp/e.txt | cut -c1-330; fi
  done
  echo "$1: in sync $ok, not $bad"
}
run default '{}'
run gloo_allreduce '{"native_allreduce": false}'
run adam_then_pack '{"adam_writes_planes": false}'
