cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o v -- $GRAFT_REPO_ROOT/tools/exp/gae_variants pmc > /dev/null 2>&1
  ls $OUT/$C
done
python - <<'PY'
import csv, os, collections, glob
root = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/pmc')
for C in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob(os.path.join(root, C, '*counter_collection.csv'))
    if not files:
        print(C, 'no counter file', os.listdir(os.path.join(root, C))); continue
    rows = list(csv.DictReader(open(files[0])))
    print(C, 'columns', list(rows[0].keys())[:20])
    agg = collections.defaultdict(list)
    for r in rows:
        if r.get('Counter_Name') == C:
            agg[r['Kernel_Name'][:50]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f'  {k:50s} n={len(v)} mean={sum(v)/len(v):.1f} min={min(v):.1f} max={max(v):.1f}')
PY
