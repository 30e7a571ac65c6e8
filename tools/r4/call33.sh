#!/bin/bash
# fused LayerNorm + UVQK projection: preload as a separate step instantiation
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/lnl_variants9.txt; : > $OUT
export HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_lnl_pre1ah2.so
timeout 300 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in "" lnl_ah2 lnl_pre1ah2 lnl_pre1; do
  if [ -z "$v" ]; then unset HSTU_HIP_LIBRARY; else export HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_$v.so; fi
  timeout 120 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a $OUT
done; done
