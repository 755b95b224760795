cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_agent
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o humanoid -- python $GRAFT_REPO_ROOT/tools/smoke_agent.py humanoid_65536 > $OUT/log.txt 2>&1
tail -5 $OUT/log.txt
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/humanoid_kernel_trace.csv 40 > $OUT/summary.txt
cat $OUT/summary.txt
rm -f $OUT/humanoid_kernel_trace.csv
