#!/bin/bash
# round 5, call 23: the yardstick probe again (call 22 lost its buffered output when the per-layer engine's library GEMM aborted
# inside a graph capture): unbuffered, the per-layer run eager and last
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c23; rm -rf $OUT; mkdir -p $OUT
timeout 400 python -u tools/exp/yardstick_probe.py 8192 4096 2>&1 | grep -v "^$" | cut -c1-700 | tee $OUT/yardstick_rank.txt | tail -30
