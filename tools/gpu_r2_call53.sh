set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c53
mkdir -p $OUT
python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -6 | tee $OUT/tests.log
python tools/rank_shapes.py loss_in_backward=0,1 worlds=8 2>&1 | grep world | tee $OUT/rank_shapes.log
python tools/rank_shapes.py loss_in_backward=0,1 worlds=8 2>&1 | grep world | tee -a $OUT/rank_shapes.log
python bench.py --workload ant --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-260
