set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c14
mkdir -p $OUT
python -m pytest tests/test_agent_gpu.py -m gpu -q --timeout 900 -k "not two_rank and not ipc" 2>&1 | tail -30 | tee $OUT/agent_tests.log
python __graft_entry__.py smoke 2>&1 | tail -4 | tee $OUT/smoke.log
