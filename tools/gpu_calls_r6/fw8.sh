cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_mlp_chain_gpu.py tests/test_headline_gpu.py tests/test_ops_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -6
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('roofline_fwd',{}).get('avg_launch_us'), d.get('roofline_fwd_infer',{}).get('avg_launch_us'), d.get('roofline_bwd',{}).get('avg_launch_us'))"
