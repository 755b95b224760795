set -x
timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 1024 --groups 2 --cold-x 24 --reps 96 2>&1 | grep "forward"
timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 1024 --groups 2 --cold-x 24 --reps 96 2>&1 | grep "forward"
