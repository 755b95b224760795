"""Summarise a rocprofv3 --kernel-trace CSV into a short per-kernel table (name truncated)."""
import csv, sys, collections, re
path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: [0, 0, 10**18, 0])
for r in rows:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    name = r['Kernel_Name']
    name = re.sub(r'\(.*', '', name)
    name = re.sub(r'<.*', '', name) if len(name) > 90 else name
    a = agg[name[:100]]
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
total = sum(a[1] for a in agg.values())
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
print(f'# kernels: {len(rows)} dispatches, busy {total/1e6:.2f} ms over a {(t1-t0)/1e6:.2f} ms window')
print(f'{"calls":>7} {"total_ms":>10} {"avg_us":>9} {"min_us":>9} {"max_us":>9} {"%":>6}  name')
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f'{a[0]:7d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:9.2f} {a[2]/1e3:9.2f} {a[3]/1e3:9.2f} {100*a[1]/total:6.2f}  {name}')
