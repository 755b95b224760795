#!/bin/bash
# round 5, call 22: the four tests that failed in call 21, fixed (2-minibatch capture failure, Dynamo-opaque model forward,
# save_best_after in the numpy-vecenv Runner test) + data for the parity yardstick at a rank's shape (tools/exp/yardstick_probe.py)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c22; rm -rf $OUT; mkdir -p $OUT
timeout 300 python -m pytest tests/test_runner_gpu.py "tests/test_agent_gpu.py::test_a_failed_capture_leaves_nothing_marked_as_packed" -m gpu -q -p no:cacheprovider --timeout 200 --tb=short -rf 2>&1 | tail -40 | cut -c1-300 | tee $OUT/pytest.txt
timeout 400 python tools/exp/yardstick_probe.py 8192 4096 2>&1 | grep -v "^$" | tail -30 | cut -c1-700 | tee $OUT/yardstick_rank.txt
