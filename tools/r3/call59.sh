#!/bin/bash
mkdir -p gpurun_out/r3
timeout 1200 python tools/fuzz_attention.py --bias --cases 200 --seed 3 2>&1 | grep -v amdgpu | tee gpurun_out/r3/fuzz59.txt | tail -30
