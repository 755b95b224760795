cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pdw -o d -- python $GRAFT_REPO_ROOT/tools/bench_dw_mfma.py 32768 256 > /tmp/pdw_log.txt 2>&1
tail -6 /tmp/pdw_log.txt
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('/tmp/pdw/d_kernel_trace.csv')))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'][:60]
    if 'mlp_dw' in n:
        agg[(n, r.get('Grid_Size', r.get('Grid_Size_X', '?')))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(agg.items()):
    v.sort()
    print(k, 'n', len(v), 'median us %.1f  min %.1f' % (v[len(v)//2] / 1e3, v[0] / 1e3))
PY
