#!/bin/bash
# round 5, call 1: where do two ranks sharing a GPU diverge?  (instrumented library, tools/exp/adam_trace_probe.py)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c1; mkdir -p $OUT
export RLG_TEST_SINGLE_GPU=1 RLG_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/_build/trace/lib.so
p=31000
run() {  # label, world, runs, env...
  label=$1; w=$2; n=$3; shift 3
  echo "=== $label (world $w)" | tee -a $OUT/trace.txt
  for i in $(seq 1 $n); do
    p=$((p+1))
    env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $p tools/exp/adam_trace_probe.py 3 2>&1 | grep -E "^RESULT|^  |Error|error" | cut -c1-400 | tee -a $OUT/trace.txt
  done
}
run rowpt_coherent 2 4 RLG_ADAM_PACK_ROWPT=1 PROBE_FLAGS=1
run rowpt_light 2 4 RLG_ADAM_PACK_ROWPT=1 PROBE_FLAGS=0
run block4x4 2 2 PROBE_FLAGS=0
run frags_world4 4 5 PROBE_FLAGS=0 RLG_BENCH_CONFIG='{"adam_frags_multi_gpu": true}'
run frags_world2_small 2 4 PROBE_FLAGS=0 PROBE_ENVS=16384 PROBE_MB=8192 RLG_BENCH_CONFIG='{"adam_frags_multi_gpu": true}'
