cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest "tests/test_agent_gpu.py::test_two_rank_bench_on_one_gpu" -m gpu -q 2>&1 | tail -1; done
