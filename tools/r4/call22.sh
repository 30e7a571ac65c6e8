#!/bin/bash
# fused LayerNorm + UVQK projection with the memory instructions spread over the chain: tests, bench, timeline
mkdir -p gpurun_out/r4
timeout 300 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -3
timeout 200 python tools/bench_ln_linear.py 2>&1 | tail -1 | tee gpurun_out/r4/lnl_bench_v2.txt
timeout 200 python tools/bench_ln_linear.py --fused-only 2>&1 | tail -1 | tee -a gpurun_out/r4/lnl_bench_v2.txt
timeout 200 python tools/trace_ln_linear.py lnl_trace > gpurun_out/r4/lnl_trace_v2.txt 2>&1; sed -n 20,50p gpurun_out/r4/lnl_trace_v2.txt; awk '/--- wave 4/,0' gpurun_out/r4/lnl_trace_v2.txt | sed -n 20,32p
