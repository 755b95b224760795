cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest "tests/test_agent_gpu.py::test_lstm_update_matches_reference_epoch" -m gpu -q -x 2>&1 | tail -40 | cut -c1-250
RLG_CHAIN_LEAN=0 timeout 600 python -m pytest "tests/test_agent_gpu.py::test_lstm_update_matches_reference_epoch" -m gpu -q -x 2>&1 | tail -3
