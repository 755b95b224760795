#!/bin/bash
# round 5, call 25: measurement only (the tree is the one call 24 verified + docs) - rocprofv3 kernel-trace summary of a rank of 8's
# epochs (tools/rank_shapes.py worlds=8), and this round's PMC measurement of the in-epoch GAE launch's HBM traffic
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c25; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof8 -o r -- python $GRAFT_REPO_ROOT/tools/rank_shapes.py worlds=8 > $OUT/prof_log8.txt 2>&1 )
python tools/prof_summary.py $OUT/prof8/r_kernel_trace.csv 30 > $OUT/prof_summary_world8.txt 2>&1; rm -rf $OUT/prof8
head -14 $OUT/prof_summary_world8.txt | cut -c1-160
timeout 240 bash tools/gpu_pmc_bench_gae.sh 2>&1 | tail -4 | cut -c1-600
cp gpurun_out/gae_pmc_traffic.json $OUT/ 2>/dev/null
