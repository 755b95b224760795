set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call8
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_headline_gpu.py -m gpu -q -x > $OUT/pytest_headline.txt 2>&1; tail -30 $OUT/pytest_headline.txt | cut -c1-300
export RLG_TEST_SINGLE_GPU=1
for i in 1 2 3 4 5 6 7 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) bench.py --gpus 2 --steps 1 --warmup 2 > $OUT/two_rank_$i.txt 2>&1
  grep '^{' $OUT/two_rank_$i.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('run $i in_sync', c.get('ranks_in_sync'), 'finite', c.get('params_finite'), c.get('allreduce'), 'ms', round(d['ms_per_step'], 1))" || tail -5 $OUT/two_rank_$i.txt
  grep "parameter probe" $OUT/two_rank_$i.txt
done
