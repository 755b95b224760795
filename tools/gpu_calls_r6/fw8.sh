cd $GRAFT_REPO_ROOT
STEPS=10 bash tools/bench_ab.sh "ship:" "h2:RLG_HIP_LIB=tools/exp/_build/h2.so" "ship2:" "h2b:RLG_HIP_LIB=tools/exp/_build/h2.so" "ship3:" "h2c:RLG_HIP_LIB=tools/exp/_build/h2.so"
