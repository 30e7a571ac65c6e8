#!/bin/bash
# round 4, GPU call 7: wide backward v2 with the DMA chunk count fixed: parity, trace, A/B
mkdir -p gpurun_out/r4
{
echo "== debug cases"; timeout 300 python tools/r4/dbg_wide.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
echo "== pytest attention + metric shapes"
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_metric_shapes_gpu.py -q -m gpu 2>&1 | tail -8
echo "== fuzz"
timeout 600 python tools/fuzz_attention.py --cases 150 --seed 41 2>&1 | grep -i "fail\|cases\|wide" | cut -c1-250 | head -20
timeout 600 python tools/fuzz_attention.py --big --cases 10 --seed 42 2>&1 | grep -i "fail\|cases\|wide" | cut -c1-250 | head
echo "== trace"
timeout 300 python tools/trace_run.py 200 2>&1 | tail -62
echo "== A/B"
P=tests/probe
HSTU_BWD_WIDE=0 timeout 300 python tools/ab_bwd.py generative_recommenders_amd/libhstu_hip.so 2>&1 | tail -1
timeout 900 python tools/ab_bwd.py generative_recommenders_amd/libhstu_hip.so $P/libhstu_w4_nostream.so $P/libhstu_w4_p1.so $P/libhstu_w4_abl64.so $P/libhstu_w4_abl32.so $P/libhstu_w4_abl96.so $P/libhstu_w4_abl256.so $P/libhstu_w4_abl3.so $P/libhstu_w4_kv3.so 2>&1 | tail -12
timeout 300 python tools/ab_bwd.py --workload M-jag generative_recommenders_amd/libhstu_hip.so 2>&1 | tail -1
} > gpurun_out/r4/call07.txt 2>&1
tail -130 gpurun_out/r4/call07.txt
