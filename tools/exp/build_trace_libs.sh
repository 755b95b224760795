#!/bin/bash
# Diagnostic build of the C-ABI library with the optimiser launches instrumented (-DRLG_ADAM_TRACE, csrc/adam_trace.hpp):
#   tools/exp/build_trace_libs.sh   ->  tools/exp/_build/trace/lib.so   (select with RLG_HIP_LIB; tools/exp/adam_trace_probe.py)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/exp/_build/trace
CS=$ROOT/rl_games_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I$ROOT/include -DRLG_ADAM_TRACE"
mkdir -p $B
make -C $CS -j8 >/dev/null
TR="optim mlp_chain mlp_chain_bx mlp_dw"
for f in $TR; do
  /opt/rocm/bin/hipcc $FLAGS -c $CS/$f.hip -o $B/$f.o &
done
wait
OTHERS=$(for o in $CS/build/*.o; do b=$(basename $o .o); case " $TR " in *" $b "*) ;; *) echo $o;; esac; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $(for f in $TR; do echo $B/$f.o; done) -o $B/lib.so
rm -f $B/*.o
ls -la $B/lib.so
