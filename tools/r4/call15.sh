#!/bin/bash
# round 4, GPU call 15: folded backward, copy-out of the parked tile by the side-B waves behind their dQ chain
mkdir -p gpurun_out/r4
{
echo "== pytest attention + metric shapes"
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_metric_shapes_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/fuzz_attention.py --big --cases 10 --seed 45 2>&1 | grep -i "fail\|cases," | cut -c1-250 | head -5
P=tests/probe
echo "== A/B M-full"; timeout 900 python tools/ab_bwd.py generative_recommenders_amd/libhstu_hip.so $P/libhstu_f_nosplit.so 2>&1 | tail -2
echo "== A/B M-jag"; timeout 900 python tools/ab_bwd.py --workload M-jag generative_recommenders_amd/libhstu_hip.so $P/libhstu_f_nosplit.so 2>&1 | tail -2
echo "== A/B 1024 users"; timeout 900 python tools/ab_bwd.py --users 1024 --launches 40 generative_recommenders_amd/libhstu_hip.so $P/libhstu_f_nosplit.so 2>&1 | tail -2
} > gpurun_out/r4/call15.txt 2>&1
tail -20 gpurun_out/r4/call15.txt
