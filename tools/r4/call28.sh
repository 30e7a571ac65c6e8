#!/bin/bash
# fused LayerNorm + UVQK projection: what the core loop costs beyond the MFMA stream (warm clocks)
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/lnl_variants4.txt; : > $OUT
for rep in 1 2; do
for v in "" lnl_abl13 lnl_abl29 lnl_abl61 lnl_abl125 lnl_abl253 lnl_abl77 lnl_abl141; do
  if [ -z "$v" ]; then unset HSTU_HIP_LIBRARY; else export HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_$v.so; fi
  timeout 120 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a $OUT
done; done
