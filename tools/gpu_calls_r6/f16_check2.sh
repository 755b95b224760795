#!/bin/bash
# round 6: fp16 form with per-workgroup gradient maxima (no atomics) - chain / ops / headline tests, then the bench rows
timeout 1500 python -m pytest tests/test_mlp_chain_gpu.py tests/test_ops_gpu.py tests/test_headline_gpu.py -q -m gpu 2>&1 | tail -8
bash tools/gpu_calls_r6/f16_bench.sh
