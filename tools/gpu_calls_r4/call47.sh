cd $GRAFT_REPO_ROOT
export RLG_TEST_SINGLE_GPU=1 RLG_BENCH_PREFLIGHT=0
p=30200; ok=0; bad=0
for i in $(seq 1 14); do
  p=$((p+1))
  r=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --steps 1 --warmup 2 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['config'].get('ranks_in_sync'))")
  [ "$r" = "True" ] && ok=$((ok+1)) || bad=$((bad+1))
done
echo "bench --gpus 2 on one GPU: in sync $ok, not $bad"
timeout 600 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -k "adam" 2>&1 | tail -2
