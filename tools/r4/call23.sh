#!/bin/bash
# fused LayerNorm + UVQK projection: fine timeline (a stamp every 4 MFMAs), default / no stores / no W requests / neither
mkdir -p gpurun_out/r4
for v in lnl_trace lnl_trace_abl1 lnl_trace_abl4 lnl_trace_abl5; do
  timeout 200 python tools/trace_ln_linear.py $v > gpurun_out/r4/${v}_fine.txt 2>&1; echo "== $v"; awk '/--- wave 0/,/--- wave 1/' gpurun_out/r4/${v}_fine.txt | sed -n 75,110p;  awk '/--- wave 4/,0' gpurun_out/r4/${v}_fine.txt | sed -n 75,100p
done
