set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c44
mkdir -p $OUT
for c in "4 4" "2 4" "4 2" "2 2"; do set -- $c; echo "== MAXBO=$1 MAXBI=$2"; RLG_DW_MAXBO=$1 RLG_DW_MAXBI=$2 timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 1024 2048 --groups 2 2>&1 | grep "dW"; done | tee $OUT/bench_chain.log
