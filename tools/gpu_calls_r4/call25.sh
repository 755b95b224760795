# order dependence of the GPU tests: every file on its own (fresh process, fresh allocator), last file first
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c25; rm -rf $OUT; mkdir -p $OUT
for f in $(ls tests/test_*gpu*.py | sort -r); do
  echo "== $f" >> $OUT/files.txt
  timeout 900 python -m pytest $f -m gpu -q 2>&1 | tail -4 >> $OUT/files.txt
done
cat $OUT/files.txt
