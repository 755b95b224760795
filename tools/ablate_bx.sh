#!/bin/bash
# Timing-only ablations of the split-bf16 chain (csrc/mlp_chain_bx.hip, -DRLG_ABL=mask: 2 no dZ stores, 4 no H loads,
# 64 every weight-plane load from the same fragments, 128 no LDS reads of the activation planes; WRONG results).
#   tools/ablate_bx.sh build "0 2 4 64 128 198"; gpurun -- tools/ablate_bx.sh run
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/tools/exp/_build_bx
CS=$ROOT/rl_games_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I$ROOT/include"
if [ "$1" = build ]; then
  rm -rf $B; mkdir -p $B
  make -C $CS -j8 >/dev/null
  for m in $2; do
    ( /opt/rocm/bin/hipcc $FLAGS -DRLG_ABL=$m -c $CS/mlp_chain_bx.hip -o $B/bx_abl$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $CS/build/*.o | grep -v mlp_chain_bx.o) $B/bx_abl$m.o -o $B/librlg_abl$m.so ) &
  done
  wait
  ls $B/*.so
else
  OUT=$ROOT/gpurun_out/ablate_bx
  mkdir -p $OUT
  : > $OUT/ablate_bx.txt
  for lib in $(ls $B/librlg_*.so); do
    echo "=== $(basename $lib)" >> $OUT/ablate_bx.txt
    RLG_HIP_LIB=$lib timeout 300 python $ROOT/tools/exp/bx_bwd_time.py ${ROWS:-32768} 2>&1 | tail -1 >> $OUT/ablate_bx.txt; RLG_HIP_LIB=$lib timeout 300 python $ROOT/tools/exp/bx_phases.py 2>&1 | grep "u1 \|dZ1 units\|dZ2 units" | tail -5 >> $OUT/ablate_bx.txt
  done
  cat $OUT/ablate_bx.txt
fi
