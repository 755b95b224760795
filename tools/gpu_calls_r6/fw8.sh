cd $GRAFT_REPO_ROOT
RLG_HIP_LIB=tools/exp/_build/bw8.so timeout 900 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -4
bash tools/bench_ab.sh "base:" "bw8:RLG_HIP_LIB=tools/exp/_build/bw8.so" "base2:" "bw8b:RLG_HIP_LIB=tools/exp/_build/bw8.so"
