cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_rank8
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o r8 -- python $GRAFT_REPO_ROOT/tools/rank_shapes.py worlds=8 > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python - <<'PY'
import csv, os, collections, re
p = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/prof_rank8/r8_kernel_trace.csv')
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp']); t1 = int(rows[-1]['End_Timestamp'])
cut = t1 - (t1 - t0) * 0.2
sel = [r for r in rows if int(r['Start_Timestamp']) >= cut]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel)
span = int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])
print(f'steady-state window {span/1e6:.1f} ms: busy {busy/1e6:.1f} ms ({100*busy/span:.1f}%), {len(sel)} dispatches, avg kernel {busy/len(sel)/1e3:.2f} us, avg gap {(span-busy)/len(sel)/1e3:.2f} us')
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    n = re.sub(r'\(.*', '', r['Kernel_Name']); n = re.sub(r'<.*', '', n) if len(n) > 80 else n
    a = agg[n[:90]]; a[0] += 1; a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f'{a[0]:6d} {a[1]/1e6:8.2f} ms {a[1]/a[0]/1e3:7.2f} us {100*a[1]/busy:5.1f}%  {n}')
# gap histogram
gaps = []
for a, b in zip(sel[:-1], sel[1:]):
    gaps.append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
import statistics
gaps.sort()
print('gap percentiles us: p50 %.2f p90 %.2f p99 %.2f max %.1f; sum of gaps > 20us: %.1f ms' % (
    gaps[len(gaps)//2]/1e3, gaps[int(len(gaps)*.9)]/1e3, gaps[int(len(gaps)*.99)]/1e3, gaps[-1]/1e3,
    sum(g for g in gaps if g > 20000)/1e6))
PY
rm -f $OUT/r8_kernel_trace.csv
