set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c26
mkdir -p $OUT
python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -5 | tee $OUT/tests.log
python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-420 | tee $OUT/bench.json
