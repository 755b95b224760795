cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-400
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/bench_kernel_trace.csv 40 > $OUT/summary.txt
cat $OUT/summary.txt
python - <<'PY'
import csv, os
p = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/prof_bench/bench_kernel_trace.csv')
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 40% of the trace = steady state; busy fraction there
t0 = int(rows[0]['Start_Timestamp']); t1 = int(rows[-1]['End_Timestamp'])
cut = t1 - (t1 - t0) * 0.25
sel = [r for r in rows if int(r['Start_Timestamp']) >= cut]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel)
span = int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])
print(f'steady-state window {span/1e6:.1f} ms: busy {busy/1e6:.1f} ms ({100*busy/span:.1f}%), {len(sel)} dispatches, avg gap {(span-busy)/len(sel)/1e3:.2f} us')
PY
cp $OUT/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -f $OUT/bench_kernel_trace.csv
