# round 6: GAE variants timed INSIDE the epoch (bench.py's HIP events bound to the 20+ in-epoch dispatches)
cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/tools/exp/_build
timeout 300 python -m pytest tests/test_gae_gpu.py tests/test_headline_gpu.py -m gpu -q -x -k "gae or dataset_preparation" 2>&1 | tail -2
STEPS=8 tools/bench_ab.sh "scan_a:" "tail_a:RLG_HIP_LIB=$B/gae_tail.so" "scan_b:" "tail_b:RLG_HIP_LIB=$B/gae_tail.so" "scan_c:" "tail_c:RLG_HIP_LIB=$B/gae_tail.so" > /dev/null
python - <<'PY'
import json, glob
for p in sorted(glob.glob('gpurun_out/ab/*_[abc].json')):
    d = json.loads([l for l in open(p) if l.startswith('{')][-1])
    r = d['roofline']
    print(p.split('/')[-1], 'epoch ms', round(d['ms_per_step'], 2), 'GAE us', round(r['avg_launch_us'], 2), 'frac', round(r['frac'], 3), 'launches', r.get('launches'))
PY
