#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python tools/fuzz_attention.py --cases 500 --seed 2 2>&1 | grep -v amdgpu | tee gpurun_out/r3/fuzz58.txt | tail -40
