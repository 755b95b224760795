#!/bin/bash
# round 5, call 26: the whole GPU suite on the tree with the central value network on the fused chain kernels
# (central_value._ValueChain; new test_central_value_chain_gradients_equal_autograd, goldens of the real reference's CV epochs)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c26; rm -rf $OUT; mkdir -p $OUT
timeout 360 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 --tb=short -rf > $OUT/pytest_full.txt 2>&1
echo "pytest rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_full.txt | tail -30
grep -n "^E " $OUT/pytest_full.txt | head -40 | cut -c1-400
