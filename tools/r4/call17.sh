#!/bin/bash
# fused LayerNorm + UVQK projection: first parity run + timing
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_ln_linear_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r4/lnl_tests.txt
cat gpurun_out/r4/lnl_tests.txt
timeout 300 python tools/bench_ln_linear.py 2>&1 | tail -3 | tee gpurun_out/r4/lnl_bench.txt
timeout 300 python tools/bench_ln_linear.py --rows 25600 2>&1 | tail -1 | tee -a gpurun_out/r4/lnl_bench.txt
