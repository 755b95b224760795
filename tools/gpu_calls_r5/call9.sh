#!/bin/bash
# round 5, call 9: one optimiser launch repeated on identical inputs, alone and next to ANOTHER process's training job on the same
# GPU (no distributed set-up, no collective): does its result depend on the neighbour?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5c9; mkdir -p $OUT
for mode in frags pack plain; do
  echo "== $mode, alone" | tee -a $OUT/stress.txt
  timeout 120 python tools/exp/adam_stress_shared.py $mode 6 2>&1 | grep -v "^/opt" | tee -a $OUT/stress.txt
done
echo "== next to bench.py (another process)" | tee -a $OUT/stress.txt
timeout 300 python bench.py --steps 120 --warmup 2 --no-cpu-baseline --no-exact-row > /tmp/bench_bg.txt 2>&1 &
BG=$!
sleep 12
for mode in frags pack plain; do
  timeout 120 python tools/exp/adam_stress_shared.py $mode 8 2>&1 | grep -v "^/opt" | tee -a $OUT/stress.txt
done
RLG_ADAM_PACK_ROWPT=1 timeout 120 python tools/exp/adam_stress_shared.py pack 8 2>&1 | grep -v "^/opt" | tee -a $OUT/stress.txt
kill $BG 2>/dev/null; wait $BG 2>/dev/null
echo "== next to the ant workload (launch-bound neighbour)" | tee -a $OUT/stress.txt
timeout 300 python bench.py --workload ant --steps 4000 --warmup 2 --no-cpu-baseline > /tmp/bench_bg.txt 2>&1 &
BG=$!
sleep 12
for mode in frags plain; do
  timeout 120 python tools/exp/adam_stress_shared.py $mode 8 2>&1 | grep -v "^/opt" | tee -a $OUT/stress.txt
done
kill $BG 2>/dev/null; wait $BG 2>/dev/null
tail -2 /tmp/bench_bg.txt | cut -c1-300
