#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_fuzz_gpu.py -q -k "ln_linear" 2>&1 | tail -12 | cut -c1-500
timeout 900 python tools/fuzz_ops.py 2>&1 | tail -6 | tee gpurun_out/r4/fuzz_ops.txt
