#!/bin/bash
# round 6: the discrete agent's trunks on the fused chain kernels (chain_net.ChainNet) - discrete + central value tests
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_discrete_gpu.py tests/test_agent_gpu.py -q -m gpu -k "discrete or central_value or two_rank" 2>&1 | tail -40
