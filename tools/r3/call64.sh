#!/bin/bash
mkdir -p gpurun_out/r3
timeout 1500 python tools/fuzz_attention.py --big --cases 40 --seed 5 2>&1 | grep -v amdgpu | tee gpurun_out/r3/fuzz_big64.txt | grep "FAIL\|EXCEPTION\|cases," | cut -c1-300
