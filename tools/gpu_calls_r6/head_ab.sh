cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_headline_gpu.py -m gpu -q -x -k "policy_head or rollout" 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error" | tail -5
bash tools/bench_ab.sh "direct:RLG_ROLLOUT_HEAD_LDS=0" "staged:" "direct2:RLG_ROLLOUT_HEAD_LDS=0" "staged2:"
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do RLG_ROLLOUT_HEAD_LDS=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph$m -o v -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; grep -h "rollout_policy_head" /tmp/ph$m/*kernel_stats.csv | cut -c1-160; done
