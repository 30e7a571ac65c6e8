#!/bin/bash
# fused LayerNorm + UVQK projection: x as whole lines through the staging (against fragment-shaped loads), nt stores
mkdir -p gpurun_out/r4
timeout 300 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -3
OUT=gpurun_out/r4/lnl_variants7.txt; : > $OUT
for rep in 1 2; do
for v in "" lnl_xfrag; do
  if [ -z "$v" ]; then unset HSTU_HIP_LIBRARY; else export HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_$v.so; fi
  timeout 120 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a $OUT
done; done
unset HSTU_HIP_LIBRARY
timeout 200 python tools/bench_ln_linear.py 2>&1 | tail -1 | tee -a $OUT
