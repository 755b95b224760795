set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call7
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export RLG_TEST_SINGLE_GPU=1
run2() { # name, extra env assignments...
  local name=$1; shift
  env "$@" timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) bench.py --gpus 2 --steps 1 --warmup 2 > $OUT/two_rank_$name.txt 2>&1
  grep '^{' $OUT/two_rank_$name.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('$name', 'in_sync', c.get('ranks_in_sync'), c.get('allreduce'), c.get('ipc_self_test'), 'ms', round(d['ms_per_step'], 1), json.dumps(c.get('collective_preflight'))[:600])" || tail -5 $OUT/two_rank_$name.txt
}
run2 default X=1
run2 no_preflight RLG_BENCH_PREFLIGHT=0
run2 no_adam_pack RLG_BENCH_PREFLIGHT=0 RLG_BENCH_CONFIG='{"adam_writes_planes": false}'
run2 rccl_fallback RLG_BENCH_PREFLIGHT=0 RLG_BENCH_CONFIG='{"native_allreduce": false}'
