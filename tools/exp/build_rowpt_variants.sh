#!/bin/bash
# Diagnosis builds of the row-per-thread Adam + planes launch (csrc/mlp_chain_bx.hip, RLG_ROWPT_VARIANT):
#   tools/exp/build_rowpt_variants.sh "1 2 3"  ->  tools/exp/_build/rowpt<N>/lib.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CS=$ROOT/rl_games_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I$ROOT/include"
make -C $CS -j8 >/dev/null
for v in $1; do
  B=$ROOT/tools/exp/_build/rowpt$v; mkdir -p $B
  X=""; V=$v
  # variant 4: the product source (variant 0) compiled without the SLP vectoriser - no packed-f32 arithmetic in the Adam body
  if [ $v = 4 ]; then X="-fno-slp-vectorize"; V=0; fi
  ( /opt/rocm/bin/hipcc $FLAGS $X -DRLG_ROWPT_VARIANT=$V -c $CS/mlp_chain_bx.hip -o $B/bx.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $CS/build/*.o | grep -v mlp_chain_bx.o) $B/bx.o -o $B/lib.so && rm $B/bx.o ) &
done
wait
ls -la $ROOT/tools/exp/_build/rowpt*/lib.so
