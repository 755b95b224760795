set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c13
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py tests/test_ops_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -3 | tee $OUT/tests.log
timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --groups 2 --dw-blocks 1024 2048 2>&1 | grep -v "forward\|backward" | tee $OUT/bench_dw.log
cd /tmp && export TMPDIR=/tmp
