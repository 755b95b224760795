set -x
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8
python -c "import torch; print(torch.cuda.get_device_name(0), torch.version.hip)"
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python __graft_entry__.py smoke 2>&1 | tail -3
python tools/bench_gae.py --rotate 1 2>&1 | tail -4
python tools/bench_gae.py --rotate 16 2>&1 | tail -4
python tools/bench_gae.py --envs 4096 --horizon 16 --rotate 1 2>&1 | tail -4
mkdir -p gpurun_out/prof1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -o gae -- python $GRAFT_REPO_ROOT/tools/bench_gae.py --rotate 16 --iters 100 2>&1 | tail -5
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof1 | head -20
