cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4c26; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
timeout 300 python tools/bench_mlp_chain.py --rows 32768 --no-lib --dw-blocks 96 128 192 256 320 384 512 768 --groups 4 --reps 20 2>/dev/null | grep -E "dW" >> $OUT/dw.txt
done
cat $OUT/dw.txt
