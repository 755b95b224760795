set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c43
mkdir -p $OUT
RLG_DW_PAIR=2 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "dw or engine" 2>&1 | tail -2 | tee $OUT/tests.log
for c in 0 2 0 2; do echo "== RLG_DW_PAIR=$c"; RLG_DW_PAIR=$c timeout 300 python tools/bench_mlp_chain.py --rows 32768 4096 --no-lib --dw-blocks 1024 --groups 2 2>&1 | grep "dW\|bias col"; done | tee $OUT/bench_chain.log
