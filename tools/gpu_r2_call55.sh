OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c55
mkdir -p $OUT
python -m pytest tests/test_agent_gpu.py -m gpu -q -x --timeout 900 > $OUT/full.log 2>&1
grep -n "^E \|Error\|FAILED\|passed\|failed" $OUT/full.log | head -20
