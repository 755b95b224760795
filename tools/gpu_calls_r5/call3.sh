#!/bin/bash
# round 5, call 3: bench.py --gpus 2 (the job that desyncs 2 of 8 with the row-per-thread Adam + planes launch) on the
# instrumented library, rows compared behind the run
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c3; mkdir -p $OUT
export RLG_TEST_SINGLE_GPU=1 RLG_BENCH_PREFLIGHT=0 RLG_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/_build/trace/lib.so RLG_ADAM_PACK_ROWPT=1
p=31200
for fl in 2 3; do
  for i in $(seq 1 10); do
    p=$((p+1))
    RLG_BENCH_ADAM_TRACE=$fl timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p bench.py --gpus 2 --steps 1 --warmup 2 > /tmp/o.txt 2> /tmp/e.txt
    grep -E "^TRACE|^  |parameter probe" /tmp/e.txt | cut -c1-420 | tee -a $OUT/trace.txt
  done
done
