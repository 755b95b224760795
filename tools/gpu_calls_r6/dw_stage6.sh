cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_mlp_chain_gpu.py -m gpu -x -q -k "dw or weight or non_finite or full_size" 2>&1 | tail -5
echo "--- staged"; timeout 120 python tools/exp/dw_bf16_check.py --reps 50 2>&1 | grep -v amdgpu.ids | tail -5
for B in 384 512; do echo "--- staged, RLG_DW_BLOCKS=$B"; RLG_DW_BLOCKS=$B timeout 120 python tools/exp/dw_bf16_check.py --reps 50 2>&1 | tail -1; done
echo "--- register form"; RLG_DW_STAGE=0 timeout 120 python tools/exp/dw_bf16_check.py --reps 50 2>&1 | tail -1
bash tools/gpu_calls_r6/dw_stage3.sh
