cd $GRAFT_REPO_ROOT
export RLG_TEST_SINGLE_GPU=1
p=30100
for i in 1 2 3 4 5 6; do
  p=$((p+1))
  PROBE_SYNC=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p tools/exp/two_rank_planes_probe.py 3 2>&1 | grep -E "^epoch|^   " | tail -6 | cut -c1-600
done
