#!/bin/bash
# fused LayerNorm + UVQK projection: two four-wave workgroups per CU (HSTU_LNL_SPLIT=1) against one of eight
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/lnl_split.txt; : > $OUT
HSTU_LNL_SPLIT=1 timeout 300 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -3 | tee -a $OUT
for rep in 1 2; do
  timeout 120 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a $OUT
  HSTU_LNL_SPLIT=1 timeout 120 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a $OUT
done
