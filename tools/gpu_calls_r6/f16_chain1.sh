#!/bin/bash
# round 6: chain kernels on fp16 x 3 products - tests of the split kernels, then the bench line
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py -q -m gpu -x -k "split or planes or non_finite or adam_step_pack or full_size or rescaling" 2>&1 | tail -30
RLG_DW_F16=0 timeout 600 python bench.py --no-cpu-baseline --no-exact-row > gpurun_out/bench_f16c.json 2> gpurun_out/bench_f16c.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_f16c.json').read().strip().splitlines()[-1])
print('ms', d['ms_per_step'], 'fwd', d['roofline_fwd']['avg_launch_us'], 'bwd', d['roofline_bwd']['avg_launch_us'], 'dw', d['roofline_mfma']['avg_launch_us'], 'infer', d['roofline_fwd_infer']['avg_launch_us'])
PY
tail -3 gpurun_out/bench_f16c.err
