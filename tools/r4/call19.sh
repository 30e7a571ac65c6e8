#!/bin/bash
# fused LayerNorm + UVQK projection: counted vmcnt (stores get another step) against draining every step; ablations
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/lnl_variants2.txt; : > $OUT
timeout 300 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in "" lnl_drain lnl_abl1 lnl_abl2 lnl_abl4 lnl_abl8; do
  if [ -z "$v" ]; then unset HSTU_HIP_LIBRARY; else export HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_$v.so; fi
  timeout 120 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a $OUT
done; done
unset HSTU_HIP_LIBRARY
timeout 200 python tools/bench_ln_linear.py 2>&1 | tail -1 | tee -a $OUT
