#!/bin/bash
# round 6: chain kernels on fp16 x 3 products - the chain tests without -x, the kernel summary of one bench run
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_mlp_chain_gpu.py -q -m gpu 2>&1 | tail -40
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_f16; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_f16 -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact-row > $GRAFT_REPO_ROOT/gpurun_out/bench_f16c2.json 2>/dev/null
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/prof_f16 -name "*kernel_trace.csv" | head -1) 2>/dev/null | head -14
