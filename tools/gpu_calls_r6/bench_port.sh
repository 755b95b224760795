#!/bin/bash
# round 6: default bench line after the cpu_baseline leg became port-only
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_port.json 2> gpurun_out/bench_port.err
echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_port.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['cpu_baseline']['kind'], d['cpu_baseline']['value'], d['gpu_over_cpu'], d['roofline']['frac'])
PY
