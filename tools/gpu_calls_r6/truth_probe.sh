cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/tools/exp/_build
for shape in "8192 4096" "65536 32768"; do
  python tools/exp/truth_probe.py $shape 2>&1 | grep -v amdgpu.ids
  RLG_HIP_LIB=$B/rowsum32.so python tools/exp/truth_probe.py $shape 2>&1 | grep -v amdgpu.ids
done
