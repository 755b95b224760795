set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c11
mkdir -p $OUT
timeout 600 python -m pytest tests/test_agent_gpu.py -m gpu -q -x --timeout 600 -k "ipc_allreduce or two_rank" 2>&1 | tail -15 | tee $OUT/ipc_tests.log
