#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_research_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 1200 python tools/fuzz_attention.py --bias --cases 300 --seed 3 2>&1 | grep -v amdgpu | tee gpurun_out/r3/fuzz61.txt | grep "FAIL\|EXCEPTION\|cases,"
