#!/bin/bash
mkdir -p gpurun_out/r4
timeout 2400 python tools/ref_triton_compare.py > gpurun_out/r4/call09_triton.txt 2>&1
grep -v "amdgpu.ids\|kpack is deprecated\|warnings.warn" gpurun_out/r4/call09_triton.txt | tail -60 | cut -c1-400
