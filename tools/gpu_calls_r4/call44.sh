cd $GRAFT_REPO_ROOT
export RLG_TEST_SINGLE_GPU=1
for i in 1 2 3 4; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29900+i)) tools/exp/two_rank_planes_probe.py 3 2>&1 | grep -E "^epoch|Error|error" | head -5
echo --
done
