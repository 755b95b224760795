set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c39
mkdir -p $OUT
RLG_TEST_SINGLE_GPU=1 python -m pytest tests/test_agent_gpu.py -m gpu -q -x --timeout 900 -k "ipc or two_rank" 2>&1 | tail -8 | tee $OUT/tests.log
RLG_TEST_SINGLE_GPU=1 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -5 | tee -a $OUT/tests.log
