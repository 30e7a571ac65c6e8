#!/bin/bash
mkdir -p gpurun_out/r4
timeout 300 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -3
timeout 200 python tools/bench_ln_linear.py 2>&1 | tail -1 | tee gpurun_out/r4/lnl_bench_v4.txt
