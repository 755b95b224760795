#!/bin/bash
# Multi-GPU rows of bench.py on ONE node, back to back: N = 1 2 4 8 (those that fit the visible GPUs), each with
# the in-graph hipIpc all-reduce (default), its two-phase variant, and RCCL.  One JSON line per row on stdout and in
# gpurun_out/scale/.  Never hangs: every run is under `timeout` and bench.py carries its own watchdog.
#     tools/scale_check.sh [steps] [warmup]
cd "$(dirname "$0")/.."
STEPS=${1:-10}; WARM=${2:-3}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/scale; mkdir -p $OUT
NG=$(python -c 'import torch; print(torch.cuda.device_count())')
PORT=29610
for N in 1 2 4 8; do
  [ "$N" -gt "$NG" ] && continue
  for MODE in ipc ipc2 rccl; do
    [ "$N" = 1 ] && [ "$MODE" != ipc ] && continue
    case $MODE in
      ipc)  CFG='{}';;
      ipc2) CFG='{"native_allreduce_two_phase": true}';;
      rccl) CFG='{"native_allreduce": false}';;
    esac
    PORT=$((PORT + 1))
    if [ "$N" = 1 ]; then
      CMD="python bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline"
    else
      CMD="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --steps $STEPS --warmup $WARM"
    fi
    echo "# N=$N $MODE" >&2
    RLG_BENCH_CONFIG="$CFG" timeout 1700 $CMD > $OUT/n${N}_$MODE.json 2> $OUT/n${N}_$MODE.err
    echo "rc=$? $(tail -1 $OUT/n${N}_$MODE.json | cut -c1-400)"
  done
done
