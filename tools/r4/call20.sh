#!/bin/bash
# fused LayerNorm + UVQK projection: s_memtime timeline of one workgroup, then rocprofv3 stats + PMC of the bench tool
mkdir -p gpurun_out/r4
timeout 200 python tools/trace_ln_linear.py > gpurun_out/r4/lnl_trace.txt 2>&1; tail -12 gpurun_out/r4/lnl_trace.txt
timeout 900 bash tools/prof_cmd.sh r04_lnl python tools/bench_ln_linear.py --iters 10 > gpurun_out/r4/lnl_prof.log 2>&1
tail -120 gpurun_out/prof_r04_lnl/summary.md
