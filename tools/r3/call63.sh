#!/bin/bash
mkdir -p gpurun_out/r3
timeout 1200 python tools/fuzz_ops.py --cases 200 --seed 4 2>&1 | grep -v amdgpu | tee gpurun_out/r3/fuzz_ops63.txt | tail -30 | cut -c1-300
