set -x
for k in 0 2 0 2; do echo "== RLG_CHAIN_WT=$k"; RLG_CHAIN_WT=$k timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 1024 --groups 2 2>&1 | grep "forward"; done
