#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab_norm.py tests/probe/libhstu_w1.so tests/probe/libhstu_w5.so 2>&1 | grep -v amdgpu | tee gpurun_out/r3/ab36.txt
