#!/bin/bash
# Diagnosis build: the translation units that hold optimiser launches compiled WITHOUT the SLP vectoriser (no compiler-formed
# v_pk_*_f32 in the Adam arithmetic) -> tools/exp/_build/noslp/lib.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/exp/_build/noslp
CS=$ROOT/rl_games_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I$ROOT/include -fno-slp-vectorize"
mkdir -p $B
make -C $CS -j8 >/dev/null
TR="optim mlp_chain mlp_chain_bx"
for f in $TR; do
  /opt/rocm/bin/hipcc $FLAGS -c $CS/$f.hip -o $B/$f.o &
done
wait
OTHERS=$(for o in $CS/build/*.o; do b=$(basename $o .o); case " $TR " in *" $b "*) ;; *) echo $o;; esac; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $(for f in $TR; do echo $B/$f.o; done) -o $B/lib.so
rm -f $B/*.o
ls -la $B/lib.so
