#!/bin/bash
# fused LayerNorm + UVQK projection inside the STU layer: tests, then the bench's layer leg with and without it
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_ln_linear_gpu.py tests/test_compute_gpu.py tests/test_metric_shapes_gpu.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --no-extra --no-cpu > gpurun_out/r4/bench_lnl_on.json 2> gpurun_out/r4/bench_lnl_on.err; tail -c 2500 gpurun_out/r4/bench_lnl_on.json
HSTU_LN_LINEAR=0 timeout 600 python bench.py --no-extra --no-cpu > gpurun_out/r4/bench_lnl_off.json 2> gpurun_out/r4/bench_lnl_off.err; tail -c 2500 gpurun_out/r4/bench_lnl_off.json
