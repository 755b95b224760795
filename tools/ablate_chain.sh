#!/bin/bash
# Timing-only ablations of the fused MLP chain kernels (csrc/mlp_chain.hip, -DRLG_ABL=mask; the masks are listed
# there).  Builds one library per mask HERE (hipcc cross-compiles), then - on the GPU box - times each with
# tools/bench_mlp_chain.py:   tools/ablate_chain.sh build "0 1 2 4 8 16 32 64 60"; gpurun -- tools/ablate_chain.sh run
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/tools/exp/_build
CS=$ROOT/rl_games_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I$ROOT/include"
if [ "$1" = build ]; then
  mkdir -p $B
  make -C $CS -j8 >/dev/null
  for m in $2; do
    ( /opt/rocm/bin/hipcc $FLAGS -DRLG_ABL=$m $EXTRA -c $CS/mlp_chain.hip -o $B/mlp_chain_abl$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $CS/build/*.o | grep -v mlp_chain.o) $B/mlp_chain_abl$m.o -o $B/librlg_abl$m.so ) &
  done
  wait
  ls -la $B/*.so
else
  OUT=$ROOT/gpurun_out/ablate
  mkdir -p $OUT
  : > $OUT/ablate.txt
  for lib in $(ls $B/librlg_*.so); do
    echo "=== $(basename $lib)" >> $OUT/ablate.txt
    RLG_HIP_LIB=$lib timeout 300 python $ROOT/tools/bench_mlp_chain.py --rows ${ROWS:-32768} --no-lib --dw-blocks 256 --groups ${GROUPS_:-2 4} --reps 30 2>&1 | grep -v "dW\|bias column" >> $OUT/ablate.txt
  done
  cat $OUT/ablate.txt
fi
