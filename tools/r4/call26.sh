#!/bin/bash
# fused LayerNorm + UVQK projection: ablations at warm clocks (60 launches before the timed 20)
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/lnl_variants3.txt; : > $OUT
for rep in 1 2; do
for v in "" lnl_abl1 lnl_abl2 lnl_abl4 lnl_abl5 lnl_abl8 lnl_abl13 lnl_ah2 lnl_ah8 lnl_drain; do
  if [ -z "$v" ]; then unset HSTU_HIP_LIBRARY; else export HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_$v.so; fi
  timeout 120 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a $OUT
done; done
