#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python tools/fuzz_attention.py --bias --cases 300 --seed 11 2>&1 | grep "FAIL\|EXCEPTION\|cases," | cut -c1-300
timeout 900 python tools/fuzz_attention.py --cases 400 --seed 12 2>&1 | grep "FAIL\|EXCEPTION\|cases," | cut -c1-300
timeout 900 python tools/fuzz_ops.py --cases 150 --seed 13 2>&1 | grep "FAIL\|EXCEPTION\|cases" | cut -c1-300
