# round 6, call 1: bf16 co-execution probe + its SQ_VALU_MFMA_COEXEC_CYCLES pass + baseline bench of the unchanged tree
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_coexec
rm -rf $OUT; mkdir -p $OUT
P=$GRAFT_REPO_ROOT/tools/exp/_build/coexec_probe_bf16
timeout 300 $P 4000 > $OUT/probe.txt 2>&1
cat $OUT/probe.txt
for C in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o v -- $P 500 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r6_coexec')
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        agg.setdefault(k, collections.OrderedDict()).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
with open(out + '/pmc_summary.txt', 'w') as fh:
    for k, c in agg.items():
        fh.write(k[:110] + '\n   ' + '  '.join(f'{n}={sum(v)/len(v):.0f}' for n, v in c.items()) + '\n')
print(open(out + '/pmc_summary.txt').read()[:6000])
PY
rm -rf $OUT/pmc_*/
cd $GRAFT_REPO_ROOT && python bench.py --no-cpu-baseline > $OUT/bench_baseline.json 2> $OUT/bench_baseline.err; tail -c 1500 $OUT/bench_baseline.json
