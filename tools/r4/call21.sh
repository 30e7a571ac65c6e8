#!/bin/bash
# fused LayerNorm + UVQK projection: s_memtime timeline of one workgroup (default, no stores, 4-tile ring)
mkdir -p gpurun_out/r4
for v in lnl_trace lnl_trace_abl1 lnl_trace_st4; do
  timeout 200 python tools/trace_ln_linear.py $v > gpurun_out/r4/$v.txt 2>&1; echo "== $v"; sed -n 1,60p gpurun_out/r4/$v.txt | head -75; tail -9 gpurun_out/r4/$v.txt
done
