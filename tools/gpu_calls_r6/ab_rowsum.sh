cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/tools/exp/_build
STEPS=6 tools/bench_ab.sh "f64:" "f32:RLG_HIP_LIB=$B/rowsum32.so" "f64b:" "f32b:RLG_HIP_LIB=$B/rowsum32.so"
