set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c16
mkdir -p $OUT
timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 1024 --groups 4 --phases 2>&1 | tee $OUT/bench_chain.log
