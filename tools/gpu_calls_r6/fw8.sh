cd $GRAFT_REPO_ROOT
RLG_HIP_LIB=tools/exp/_build/h2.so timeout 900 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed"
bash tools/bench_ab.sh "ship:" "h2:RLG_HIP_LIB=tools/exp/_build/h2.so" "h1:RLG_HIP_LIB=tools/exp/_build/h1.so" "ship2:" "h2b:RLG_HIP_LIB=tools/exp/_build/h2.so" "h1b:RLG_HIP_LIB=tools/exp/_build/h1.so"
RLG_HIP_LIB=tools/exp/_build/h2.so python tools/exp/bx_fwd_phases.py 32768 train 2>&1 | grep -v amdgpu.ids | tail -12
