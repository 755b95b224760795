python -m pytest tests/test_agent_gpu.py -m gpu -q -x --timeout 900 -k "folded_launches" 2>&1 | tail -25
