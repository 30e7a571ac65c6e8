#!/bin/bash
# round 4, GPU call 4: wide backward v2 (side work in the MFMA streams, asm-owned accumulators, direct stores): parity, trace, A/B
mkdir -p gpurun_out/r4
{
echo "== pytest attention + metric shapes (wide v2)"
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_metric_shapes_gpu.py -x -q -m gpu 2>&1 | tail -15
echo "== fuzz"
timeout 600 python tools/fuzz_attention.py --cases 150 --seed 41 2>&1 | grep -i "fail\|cases\|wide" | head -20
timeout 600 python tools/fuzz_attention.py --big --cases 10 --seed 42 2>&1 | grep -i "fail\|cases\|wide" | head
echo "== trace"
timeout 300 python tools/trace_run.py 200 2>&1 | tail -75
echo "== A/B"
P=tests/probe
HSTU_BWD_WIDE=0 timeout 300 python tools/ab_bwd.py generative_recommenders_amd/libhstu_hip.so 2>&1 | tail -1
timeout 900 python tools/ab_bwd.py generative_recommenders_amd/libhstu_hip.so $P/libhstu_w2_base.so $P/libhstu_w2_nostream.so $P/libhstu_w2_p1.so $P/libhstu_w2_abl64.so $P/libhstu_w2_abl32.so $P/libhstu_w2_abl96.so $P/libhstu_w2_abl256.so $P/libhstu_w2_abl3.so 2>&1 | tail -12
timeout 300 python tools/ab_bwd.py --workload M-jag generative_recommenders_amd/libhstu_hip.so 2>&1 | tail -1
} > gpurun_out/r4/call04.txt 2>&1
tail -150 gpurun_out/r4/call04.txt
