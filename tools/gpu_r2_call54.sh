OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c54
mkdir -p $OUT
python -m pytest tests -m gpu -q -v --timeout 900 > $OUT/full.log 2>&1
grep -n "Fatal\|fault\|Abort\|HSA\|error" $OUT/full.log | head -10
grep -n "PASSED\|FAILED" $OUT/full.log | tail -3
grep -n "File \"/tmp\|File \"/root" $OUT/full.log | head -20
