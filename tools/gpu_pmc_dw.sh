cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_dw
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o v -- python $GRAFT_REPO_ROOT/tools/bench_dw_mfma.py 32768 256 > /dev/null 2>&1
done
python - <<'PY'
import csv, os, collections, glob
root = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/pmc_dw')
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob(os.path.join(root, C, '*counter_collection.csv'))
    if not files:
        print(C, 'no counter file'); continue
    rows = list(csv.DictReader(open(files[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        if r.get('Counter_Name') == C and 'mlp_dw' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:40], r.get('Grid_Size', '?'))].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(f'{C} {k[0]:40s} grid {k[1]:>8s} n={len(v)} mean={sum(v)/len(v):.1f} KiB min={min(v):.1f} max={max(v):.1f}')
PY
rm -rf $OUT/*/v_kernel_trace.csv
