cd $GRAFT_REPO_ROOT
export RLG_TEST_SINGLE_GPU=1
for i in 1 2 3 4 5 6; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500+i)) bench.py --gpus 2 --steps 1 --warmup 2 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('in_sync', c.get('ranks_in_sync'), 'allreduce', c.get('allreduce'), 'selftest', c.get('ipc_self_test'), 'finite', c.get('params_finite'), 'ms', round(d['ms_per_step'],1), {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('self_test','us_per_allreduce')}) for k,v in (c.get('collective_check') or {}).items()})"
done
echo "--- lean off"
for i in 1 2 3 4; do
RLG_CHAIN_LEAN=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+i)) bench.py --gpus 2 --steps 1 --warmup 2 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('in_sync', c.get('ranks_in_sync'), 'allreduce', c.get('allreduce'))"
done
