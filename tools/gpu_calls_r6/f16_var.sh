#!/bin/bash
# which epoch is the slow one? default flags, then warmup 3 / steps 8
mkdir -p gpurun_out/f16
python bench.py --no-cpu-baseline --no-exact-row > gpurun_out/f16/v1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-exact-row --warmup 3 --steps 8 > gpurun_out/f16/v2.json 2>/dev/null
RLG_HIP_LIB=tools/exp/_build/bx_bf16.so RLG_DW_F16=0 python bench.py --no-cpu-baseline --no-exact-row --warmup 3 --steps 8 > gpurun_out/f16/v3.json 2>/dev/null
python - <<'PY'
import json
for n in ('v1','v2','v3'):
    d=json.loads(open(f'gpurun_out/f16/{n}.json').read().strip().splitlines()[-1])
    r=lambda k: round(d[k]['avg_launch_us'],1) if k in d else None
    print(n, round(d['ms_per_step'],2), d['ms_per_step_stats']['each'], r('roofline_fwd'), r('roofline_bwd'), r('roofline_mfma'))
PY
