# the whole GPU suite + smoke + the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6_suite
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/r6_suite/pytest_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
python bench.py > gpurun_out/r6_suite/bench.json 2> gpurun_out/r6_suite/bench.err; tail -c 600 gpurun_out/r6_suite/bench.json
