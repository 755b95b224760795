cd $GRAFT_REPO_ROOT
for w in 0 5; do echo "== wave $w of workgroup 70 (layer 1)"; RLG_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/_build/dwstamps_w$w.so python tools/exp/dw_stage_phases.py 2>&1 | grep -v amdgpu.ids; done
