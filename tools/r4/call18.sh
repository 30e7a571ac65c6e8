#!/bin/bash
# fused LayerNorm + UVQK projection: ablations and parameter variants (one process each, same box)
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/lnl_variants.txt; : > $OUT
timeout 300 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in "" lnl_st4 lnl_ah2 lnl_ah6 lnl_ah8 lnl_abl1 lnl_abl2 lnl_abl3 lnl_abl4 lnl_abl5 lnl_abl8 lnl_abl9; do
  if [ -z "$v" ]; then unset HSTU_HIP_LIBRARY; else export HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_$v.so; fi
  timeout 120 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a $OUT
done; done
