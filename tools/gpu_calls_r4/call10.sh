set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call10
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python tools/exp/parity_drift.py rank > $OUT/drift_rank.txt 2>&1; grep -v "Warn\|amdgpu.ids" $OUT/drift_rank.txt | cut -c1-330
timeout 900 python -m pytest tests/test_headline_gpu.py -m gpu -q > $OUT/pytest_headline.txt 2>&1; tail -30 $OUT/pytest_headline.txt | cut -c1-400
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-exact-row --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.load(sys.stdin); print('adam writes planes ', d['ms_per_step'])"
  RLG_BENCH_CONFIG='{"adam_writes_planes": false}' timeout 600 python bench.py --no-cpu-baseline --no-exact-row --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.load(sys.stdin); print('pack launch        ', d['ms_per_step'])"
done | tee $OUT/bench_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact-row --steps 3 --warmup 1 > $OUT/prof_log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $OUT/prof/bench_kernel_trace.csv 30 > $OUT/prof_summary.txt; cp $OUT/prof/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof; head -16 $OUT/prof_summary.txt
cd $GRAFT_REPO_ROOT
export RLG_TEST_SINGLE_GPU=1
bad=0; for i in 1 2 3 4 5 6; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) bench.py --gpus 2 --steps 1 --warmup 2 > $OUT/two_rank_$i.txt 2>&1
  grep '^{' $OUT/two_rank_$i.txt | tail -1 | grep -q '"ranks_in_sync": true' || bad=$((bad+1))
done; echo "two-rank bench with the collective check behind the run: $bad of 6 out of sync"
grep '^{' $OUT/two_rank_1.txt | tail -1 | python -c "import sys,json; d=json.load(sys.stdin); print(json.dumps(d['config'].get('collective_check'))[:500])"
