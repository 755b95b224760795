cd $GRAFT_REPO_ROOT
for wl in ant lstm; do
for v in ship w4 fw4 bw4 ship w4; do
  if [ $v = ship ]; then unset RLG_HIP_LIB; else export RLG_HIP_LIB=tools/exp/_build/$v.so; fi
  python bench.py --workload $wl --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl $v', round(d['ms_per_step'],3), d.get('ms_per_step_stats',{}).get('each'))"
done; done
