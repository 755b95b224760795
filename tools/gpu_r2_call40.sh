set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c40
mkdir -p $OUT
RLG_CHAIN_WAVES=84 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "forward" 2>&1 | tail -2 | tee $OUT/tests.log
for w in 0 84; do echo "== RLG_CHAIN_WAVES=$w"; RLG_CHAIN_WAVES=$w timeout 300 python tools/bench_mlp_chain.py --rows 32768 65536 --no-lib --dw-blocks 1024 --groups 4 2 2>&1 | grep "forward"; done | tee $OUT/bench_chain.log
