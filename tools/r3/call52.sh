#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_research_gpu.py tests/test_configs_gpu.py -m gpu -q -x 2>&1 | tail -6
timeout 300 python tools/bench_research.py books --kernels 2>&1 | grep -v amdgpu | grep "solo_bias\|fwd_ms\|bwd_ms" | tee gpurun_out/r3/research_books52.txt
