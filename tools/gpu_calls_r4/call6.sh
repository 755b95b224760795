set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call6
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -k "adam_step_pack or adam_written or three_epochs or post_step_log_val" > $OUT/pytest_new.txt 2>&1; tail -15 $OUT/pytest_new.txt
for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-exact-row --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.load(sys.stdin); print('adam writes planes ', d['ms_per_step'], d['roofline_fwd']['avg_launch_us'], d['roofline_bwd']['avg_launch_us'])"
  RLG_BENCH_CONFIG='{"adam_writes_planes": false}' timeout 600 python bench.py --no-cpu-baseline --no-exact-row --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.load(sys.stdin); print('pack launch        ', d['ms_per_step'], d['roofline_fwd']['avg_launch_us'], d['roofline_bwd']['avg_launch_us'])"
done | tee $OUT/bench_ab.txt
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt
