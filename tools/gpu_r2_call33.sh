set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c33
mkdir -p $OUT
RLG_TEST_SINGLE_GPU=1 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8 | tee $OUT/tests.log
python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-420 | tee $OUT/bench.json
