# round 6: 32-row tiles (two workgroups per CU, <= 256 registers) for the split-bf16 chain kernels - A/B inside the epoch
cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/tools/exp/_build
RLG_HIP_LIB=$B/bx_g2.so RLG_CHAIN_BWD_GROUPS=2 timeout 900 python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -4
STEPS=6 tools/bench_ab.sh "g4_g4:" "g4_bwd2:RLG_CHAIN_BWD_GROUPS=2" "fwd2_g4:RLG_HIP_LIB=$B/bx_g2.so" "fwd2_bwd2:RLG_HIP_LIB=$B/bx_g2.so,RLG_CHAIN_BWD_GROUPS=2" "g4_g4_b:"
