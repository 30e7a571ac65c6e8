#!/bin/bash
echo "== current"; timeout 600 python tools/fuzz_attention.py --bias --cases 40 --seed 3 2>&1 | grep "FAIL\|cases,"
echo "== head loop off"; HSTU_BIAS_HEAD_LOOP=0 timeout 600 python tools/fuzz_attention.py --bias --cases 40 --seed 3 2>&1 | grep "FAIL\|cases,"
echo "== no flip (bf16 bias TU)"; HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_noflipb.so timeout 600 python tools/fuzz_attention.py --bias --cases 40 --seed 3 2>&1 | grep "FAIL\|cases,"
