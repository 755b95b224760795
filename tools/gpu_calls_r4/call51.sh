cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --workload ant --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['config']['chain_products'])"
timeout 300 python bench.py --no-cpu-baseline --no-exact-row 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['config']['chain_products']['backward'][:40])"
