cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -4
bash tools/bench_ab.sh "old:RLG_HIP_LIB=tools/exp/_build/fwold.so" "noring_split:RLG_HIP_LIB=tools/exp/_build/fwnoring.so" "ring4_split:" "pairs_split:RLG_HIP_LIB=tools/exp/_build/fwpairs.so" "old2:RLG_HIP_LIB=tools/exp/_build/fwold.so" "noring_split2:RLG_HIP_LIB=tools/exp/_build/fwnoring.so" "ring4_split2:" "pairs_split2:RLG_HIP_LIB=tools/exp/_build/fwpairs.so"
python tools/exp/bx_fwd_phases.py 32768 train 2>&1 | grep -v amdgpu.ids | tail -10
