set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4call9
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export RLG_TEST_SINGLE_GPU=1
loop() { # name count env...
  local name=$1 count=$2; shift 2
  local bad=0
  for i in $(seq 1 $count); do
    env "$@" timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) bench.py --gpus 2 --steps 1 --warmup 2 > $OUT/run_${name}_$i.txt 2>&1
    if grep '^{' $OUT/run_${name}_$i.txt | tail -1 | grep -q '"ranks_in_sync": true'; then rm $OUT/run_${name}_$i.txt; else bad=$((bad+1)); fi
  done
  echo "VARIANT $name: $bad of $count runs out of sync" | tee -a $OUT/summary.txt
}
loop no_pack_no_preflight 12 RLG_BENCH_PREFLIGHT=0 RLG_BENCH_CONFIG='{"adam_writes_planes": false}'
loop pack_no_preflight 12 RLG_BENCH_PREFLIGHT=0
loop rccl_fallback 8 RLG_BENCH_PREFLIGHT=0 RLG_BENCH_CONFIG='{"native_allreduce": false}'
loop eager 8 RLG_BENCH_PREFLIGHT=0 RLG_BENCH_CONFIG='{"hip_graphs": false}'
cat $OUT/summary.txt
