#!/bin/bash
# fused LayerNorm + UVQK projection: next block's rows (whole lines) requested under the last tile; ring of two tiles
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/lnl_variants8.txt; : > $OUT
export HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_lnl_pre1.so
timeout 300 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in "" lnl_pre1 lnl_pre1ah2 lnl_st2; do
  if [ -z "$v" ]; then unset HSTU_HIP_LIBRARY; else export HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_$v.so; fi
  timeout 120 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a $OUT
done; done
