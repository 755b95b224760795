cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
RLG_CHAIN_LEAN=$v timeout 300 python bench.py --workload ant --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lean $v: ant', round(d['ms_per_step'],2), 'ms')"
done
