OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c29
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "test_chain_forward_matches_fp64 and 48" 2>&1 | tail -40 > $OUT/t1.log
RLG_TEST_SINGLE_GPU=1 python -m pytest tests/test_agent_gpu.py -m gpu -q -x --timeout 900 -k "test_update_matches_reference_epoch" 2>&1 | tail -60 > $OUT/t2.log
