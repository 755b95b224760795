cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_agent_gpu.py -m gpu -q -k "weights_set_between or graph_replays or rollout_step_graphs" 2>&1 | tail -12 | cut -c1-220
