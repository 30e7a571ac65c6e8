#!/bin/bash
# fused LayerNorm + UVQK projection: next block's rows requested under the last tile; experiments: W requested twice, stores into 1 MiB
mkdir -p gpurun_out/r4
timeout 300 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -3
OUT=gpurun_out/r4/lnl_variants5.txt; : > $OUT
for rep in 1 2; do
for v in "" lnl_nopre lnl_abl1 lnl_abl256 lnl_abl512; do
  if [ -z "$v" ]; then unset HSTU_HIP_LIBRARY; else export HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_$v.so; fi
  timeout 120 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a $OUT
done; done
