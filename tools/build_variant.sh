#!/bin/bash
# A variant build of the C-ABI library beside the product one (same-box A/B through RLG_HIP_LIB, tools/bench_ab.sh):
#   tools/build_variant.sh <name> "<extra hipcc flags>" <source> [<source> ...]     (sources without .hip)
#   -> tools/exp/_build/<name>.so  = the product's objects with the named sources recompiled under the extra flags
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2; shift 2
B=$ROOT/tools/exp/_build/$NAME.d
CS=$ROOT/rl_games_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I$ROOT/include $EXTRA"
mkdir -p $B
make -C $CS -j8 >/dev/null
for f in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -c $CS/$f.hip -o $B/$f.o &
done
wait
OTHERS=$(for o in $CS/build/*.o; do b=$(basename $o .o); case " $* " in *" $b "*) ;; *) echo $o;; esac; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $(for f in "$@"; do echo $B/$f.o; done) -o $ROOT/tools/exp/_build/$NAME.so
rm -rf $B
ls -la $ROOT/tools/exp/_build/$NAME.so
