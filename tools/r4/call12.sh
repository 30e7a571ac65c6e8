#!/bin/bash
# round 4, GPU call 12: folded backward with the 32x32x16 dQ chains: parity + A/B against the 16x16x32 phase
mkdir -p gpurun_out/r4
{
echo "== pytest attention + metric shapes"
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_metric_shapes_gpu.py -x -q -m gpu 2>&1 | tail -5
echo "== fuzz slice"
timeout 600 python tools/fuzz_attention.py --cases 150 --seed 41 2>&1 | grep -i "fail\|cases," | cut -c1-250 | head
timeout 600 python tools/fuzz_attention.py --big --cases 10 --seed 42 2>&1 | grep -i "fail\|cases," | cut -c1-250 | head
echo "== A/B M-full"
P=tests/probe
timeout 900 python tools/ab_bwd.py generative_recommenders_amd/libhstu_hip.so $P/libhstu_f_old.so $P/libhstu_f_ah1.so $P/libhstu_f_ah3.so 2>&1 | tail -4
echo "== A/B M-jag"
timeout 900 python tools/ab_bwd.py --workload M-jag generative_recommenders_amd/libhstu_hip.so $P/libhstu_f_old.so 2>&1 | tail -2
echo "== A/B 1024 users"
timeout 900 python tools/ab_bwd.py --users 1024 --launches 40 generative_recommenders_amd/libhstu_hip.so $P/libhstu_f_old.so 2>&1 | tail -2
} > gpurun_out/r4/call12.txt 2>&1
tail -40 gpurun_out/r4/call12.txt
