#!/bin/bash
# round 6: weight-gradient launch on three fp16 plane products (RLG_DW_F16=1) beside the six-bf16 form
for rep in 1 2; do
RLG_DW_F16=0 python tools/exp/dw_bf16_check.py --reps 100
RLG_DW_F16=1 python tools/exp/dw_bf16_check.py --reps 100
done
