OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c30
mkdir -p $OUT
RLG_CHAIN_WAVES=8 python tools/exp/debug_w8.py 2>&1 | grep "W=" > $OUT/dbg.log
RLG_CHAIN_WAVES=4 python tools/exp/debug_w8.py 2>&1 | grep "BAD" >> $OUT/dbg.log
python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 600 -k loss 2>&1 | tail -30 > $OUT/loss.log
