#!/bin/bash
# round 5, call 20: the whole GPU suite on the pruned tree (fused tail, adam_frags and rowpt gone; Runner-on-device tests new)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c20; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 | tee $OUT/pytest.txt
