# round 6, call 2: same-box A/B of the plane split's residual form (v_pk_add_f32 vs v_sub_f32), bench.py rows
cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/tools/exp/_build
STEPS=6 tools/bench_ab.sh "default:" "pk0_all:RLG_HIP_LIB=$B/pk0_all.so" "pk0_dw:RLG_HIP_LIB=$B/pk0_dw.so" "noslp:RLG_HIP_LIB=$B/noslp.so" "default2:" "pk0_all2:RLG_HIP_LIB=$B/pk0_all.so"
python - <<'PY'
import json, glob
for p in sorted(glob.glob('gpurun_out/ab/*.json')):
    try:
        d = json.loads([l for l in open(p) if l.startswith('{')][-1])
    except Exception as e:
        print(p, 'FAILED'); continue
    print(p.split('/')[-1], 'ms', round(d['ms_per_step'], 2), {k: round(d[k]['avg_launch_us'], 1) for k in d if k.startswith('roofline') and isinstance(d[k], dict) and 'avg_launch_us' in d[k]})
PY
