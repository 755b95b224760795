# round 6: first run of the staged weight-gradient kernel: its tests, accuracy/time tool, in-epoch A/B against the register form
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_mlp_chain_gpu.py -m gpu -x -q -k "dw or weight or non_finite or full_size" 2>&1 | tail -15
echo "--- staged"; timeout 120 python tools/exp/dw_bf16_check.py --reps 50 2>&1 | grep -v amdgpu.ids
echo "--- register form"; RLG_DW_STAGE=0 timeout 120 python tools/exp/dw_bf16_check.py --reps 50 2>&1 | grep -v amdgpu.ids
STEPS=6 tools/bench_ab.sh "stage:" "reg:RLG_DW_STAGE=0" "stage2:"
python - <<'PY'
import json, glob
for p in sorted(glob.glob('gpurun_out/ab/*.json')):
    try:
        d = json.loads([l for l in open(p) if l.startswith('{')][-1])
    except Exception as e:
        print(p, 'FAILED'); continue
    print(p.split('/')[-1], 'ms', round(d['ms_per_step'], 2), {k: round(d[k]['avg_launch_us'], 1) for k in d if k.startswith('roofline') and isinstance(d[k], dict) and 'avg_launch_us' in d[k]})
PY
