set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c45
mkdir -p $OUT
for a in elu None relu elu None relu; do echo "== act $a"; timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 1024 --groups 2 --act $a 2>&1 | grep "forward\|backward"; done | tee $OUT/bench_chain.log
