set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c23
mkdir -p $OUT
python -m pytest tests/test_mlp_chain_gpu.py -m gpu -q -x --timeout 600 -k "loss or fold or engine" 2>&1 | tail -15 | tee $OUT/tests.log
python -m pytest tests/test_headline_gpu.py tests/test_agent_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -15 | tee -a $OUT/tests.log
python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-420 | tee $OUT/bench.json
python bench.py --workload ant --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-300 | tee $OUT/bench_ant.json
