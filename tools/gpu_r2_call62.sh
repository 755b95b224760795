OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c62
mkdir -p $OUT
for v in 0 1; do
  python -m pytest tests -m gpu -q --timeout 900 > $OUT/full_$v.log 2>&1
  echo "== run $v rc=$?"; grep -n "passed\|failed\|Fatal\|^FAILED" $OUT/full_$v.log | head -8
done
