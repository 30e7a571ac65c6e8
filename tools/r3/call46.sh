#!/bin/bash
mkdir -p gpurun_out/r3
timeout 300 python tools/bench_gemm_backend.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3/gemm_backend46.txt
