#!/bin/bash
# tools/exp/build_dw_stamps.sh <workgroup> <wave>  ->  tools/exp/_build/dwstamps/lib.so  (csrc/mlp_dw.hip with -DRLG_DW_STAMPS)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
B=$ROOT/tools/exp/_build/dwstamps; mkdir -p $B
CS=$ROOT/rl_games_amd/csrc
make -C $CS -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I$ROOT/include -DRLG_DW_STAMPS=${1:-0} -DRLG_DW_STAMPS_WAVE=${2:-0} -c $CS/mlp_dw.hip -o $B/mlp_dw.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $CS/build/*.o | grep -v mlp_dw.o) $B/mlp_dw.o -o $B/lib.so
rm $B/mlp_dw.o; ls -la $B/lib.so
