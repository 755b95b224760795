#!/bin/bash
# round 6: the fp16 form end to end - default bench line twice, then ant / lstm
mkdir -p gpurun_out/f16
for i in 1 2; do python bench.py --no-cpu-baseline > gpurun_out/f16/bench$i.json 2> gpurun_out/f16/bench$i.err; done
python bench.py --workload ant --no-cpu-baseline > gpurun_out/f16/ant.json 2>/dev/null
python bench.py --workload lstm --no-cpu-baseline > gpurun_out/f16/lstm.json 2>/dev/null
python - <<'PY'
import json
for n in ('bench1','bench2','ant','lstm'):
    d=json.loads(open(f'gpurun_out/f16/{n}.json').read().strip().splitlines()[-1])
    r=lambda k: (k[9:], round(d[k]['avg_launch_us'],1), round(d[k]['frac'],3)) if k in d else None
    print(n, round(d['ms_per_step'],2), round(d['value']/1e6,2), r('roofline_fwd'), r('roofline_bwd'), r('roofline_mfma'), r('roofline_fwd_infer'), d.get('exact_products_ms_per_step'), d['config'].get('weight_gradient_products'))
PY
tail -2 gpurun_out/f16/bench1.err
