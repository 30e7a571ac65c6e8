#!/bin/bash
# fused LayerNorm + UVQK projection, y leaves as 16 rows x 64 B per store instruction (through wave-private LDS): tests, bench, timeline
mkdir -p gpurun_out/r4
timeout 300 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -3
timeout 200 python tools/bench_ln_linear.py 2>&1 | tail -1 | tee gpurun_out/r4/lnl_bench_v3.txt
timeout 200 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a gpurun_out/r4/lnl_bench_v3.txt
timeout 200 python tools/trace_ln_linear.py lnl_trace > gpurun_out/r4/lnl_trace_v3.txt 2>&1; awk '/--- wave 0/,/--- wave 1/' gpurun_out/r4/lnl_trace_v3.txt | sed -n 75,100p
