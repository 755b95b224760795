set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call11
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/exp/rollout_consistency.py > $OUT/rollout_consistency_pipe.txt 2>&1; grep -v "Warn\|amdgpu.ids" $OUT/rollout_consistency_pipe.txt | cut -c1-220
RLG_CHAIN_PIPE1=0 timeout 600 python tools/exp/rollout_consistency.py > $OUT/rollout_consistency_unit.txt 2>&1; grep -v "Warn\|amdgpu.ids" $OUT/rollout_consistency_unit.txt | cut -c1-220
timeout 600 python -m pytest tests -m gpu -q -k "adam_step_pack or adam_written" 2>&1 | tail -3
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-exact-row --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.load(sys.stdin); print('adam writes planes ', d['ms_per_step'])"
  RLG_BENCH_CONFIG='{"adam_writes_planes": false}' timeout 600 python bench.py --no-cpu-baseline --no-exact-row --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.load(sys.stdin); print('pack launch        ', d['ms_per_step'])"
done | tee $OUT/bench_ab.txt
