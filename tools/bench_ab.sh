#!/bin/bash
# A/B rows of bench.py inside ONE gpurun call (box-to-box spread is +-3 %): each argument is "label:ENV=VAL,ENV=VAL"
#     gpurun -- tools/bench_ab.sh "pipe4:RLG_CHAIN_FWD_GROUPS=4" "old2:RLG_CHAIN_PIPE=0"
cd "$(dirname "$0")/.."
OUT=gpurun_out/ab; mkdir -p $OUT
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    python bench.py --steps ${STEPS:-8} --warmup 3 --no-cpu-baseline > $OUT/$label.json 2> $OUT/$label.err )
  python - "$label" "$OUT/$label.json" <<'PY'
import json, sys
label, path = sys.argv[1:3]
try:
    d = json.loads([l for l in open(path) if l.startswith('{')][-1])
    r = lambda k: ('%s %.1f us (%.3f)' % (k[9:], d[k]['avg_launch_us'], d[k]['frac'])) if k in d else ''
    print('%-14s %.2f ms/epoch  %.3f M env-steps/s   %s  %s  %s' % (label, d['ms_per_step'], d['value'] / 1e6, r('roofline_fwd'), r('roofline_fwd_infer'), r('roofline_bwd')))
except Exception as e:
    print(label, 'FAILED', e)
PY
done
