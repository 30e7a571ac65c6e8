#!/bin/bash
# fused LayerNorm + UVQK projection: non-temporal stores of y
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/lnl_variants6.txt; : > $OUT
for rep in 1 2; do
for v in "" lnl_nt; do
  if [ -z "$v" ]; then unset HSTU_HIP_LIBRARY; else export HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_$v.so; fi
  timeout 120 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a $OUT
done; done
