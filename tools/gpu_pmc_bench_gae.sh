# HBM traffic of the in-epoch GAE launch, measured by rocprofv3 --pmc ON bench.py itself (separate
# passes per counter, --kernel-trace only).  Writes gpurun_out/gae_pmc_traffic.json in the format
# bench.py reads from profiles/gae_pmc_traffic.json.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gae_bench
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o v -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-row > $OUT/$C.log 2>&1
done
python - <<'PY'
import csv, glob, json, os
root = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/pmc_gae_bench')
vals = {}
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob(os.path.join(root, C, '**', '*counter_collection.csv'), recursive=True)
    v = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == C and 'gae_envmajor_kernel' in r['Kernel_Name']:
                v.append(float(r['Counter_Value']))
    vals[C] = v
    print(C, 'launches', len(v), 'KiB per launch', v)
if vals['FETCH_SIZE'] and vals['WRITE_SIZE']:
    fetch = sum(vals['FETCH_SIZE']) / len(vals['FETCH_SIZE']) * 1024 * 2      # gfx950: FETCH_SIZE counts 64 B per 128-B request
    write = sum(vals['WRITE_SIZE']) / len(vals['WRITE_SIZE']) * 1024
    out = {'65536x32': {'traffic_bytes': int(fetch + write), 'fetch_bytes': int(fetch), 'write_bytes': int(write),
                        'launches': len(vals['FETCH_SIZE']),
                        'note': 'rocprofv3 --pmc FETCH_SIZE (x2: gfx950 wide-read correction, MI355X_MICROARCH.md) + '
                                'WRITE_SIZE of the in-epoch GAE dispatches of `python bench.py --steps 2 --warmup 1` '
                                '(tools/gpu_pmc_bench_gae.sh), mean per launch'}}
    json.dump(out, open(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/gae_pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(out))
PY
find $OUT -name "*.csv" -delete
