#!/bin/bash
mkdir -p gpurun_out/r3

timeout 300 python tools/bench_research.py books --kernels 2>&1 | grep -v amdgpu | tee gpurun_out/r3/research_books49.txt
