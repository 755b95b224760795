set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c46
mkdir -p $OUT
for k in 0 2048 4096 8192 16384 0; do echo "== RLG_CHAIN_SKEW=$k"; RLG_CHAIN_SKEW=$k timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 1024 --groups 0 2>&1 | grep "forward\|backward"; done | tee $OUT/bench_chain.log
