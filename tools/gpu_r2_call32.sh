OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c32
mkdir -p $OUT
RLG_TEST_SINGLE_GPU=1 python -m pytest tests/test_agent_gpu.py -m gpu -q -x --timeout 900 -k "test_odd_shapes_match_oracle_epoch" 2>&1 | tail -60 > $OUT/t.log
