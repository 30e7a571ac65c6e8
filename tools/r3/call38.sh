#!/bin/bash
mkdir -p gpurun_out/r3
python tools/bench_gemm_layout.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3/gemm_layout38.txt
