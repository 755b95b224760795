#!/bin/bash
# Run of the split-bf16 weight gradients: tests, slice-count sweep, epoch time against the f32 form.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/gpu_dw_bf16.sh'
mkdir -p gpurun_out
out=gpurun_out/dw_bf16.txt
: > $out
echo "== tests (chain, dW, ops, headline)" >> $out
timeout 400 python -m pytest tests/test_mlp_chain_gpu.py tests/test_ops_gpu.py tests/test_headline_gpu.py -q -m gpu 2>&1 | tail -5 >> $out
for rows in 32768 4096; do
  for m in 1 0; do
    echo "== RLG_DW_BF16=$m rows $rows" >> $out
    RLG_DW_BF16=$m timeout 120 python tools/bench_mlp_chain.py --no-lib --groups 4 --rows $rows --reps 200 --dw-blocks 64 128 256 512 1024 2>&1 | grep "dW (" >> $out
  done
done
for cfg in "1 0" "1 128" "1 256" "1 512" "0 0"; do
  set -- $cfg
  echo "== bench RLG_DW_BF16=$1 RLG_DW_BLOCKS=$2" >> $out
  RLG_DW_BF16=$1 RLG_DW_BLOCKS=$2 timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> $out 2>&1
done
cat $out
