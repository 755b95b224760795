set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call13
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_headline_gpu.py -m gpu -q -k "three_epochs" > $OUT/pytest_drift.txt 2>&1; tail -25 $OUT/pytest_drift.txt | cut -c1-300
timeout 1800 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest.txt 2>&1; tail -30 $OUT/pytest.txt | cut -c1-250
