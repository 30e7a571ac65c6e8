#!/bin/bash
# round 4, GPU call 2: cycle trace of the wide backward + ablations (one process, HIP events)
mkdir -p gpurun_out/r4
{
echo "== trace"
timeout 300 python tools/trace_run.py 200 2>&1 | tail -80
echo "== A/B"
P=tests/probe
timeout 900 python tools/ab_bwd.py generative_recommenders_amd/libhstu_hip.so $P/libhstu_w_base.so $P/libhstu_w_kv0.so $P/libhstu_w_ah2.so $P/libhstu_w_ah6.so $P/libhstu_w_dq3.so $P/libhstu_w_abl64.so $P/libhstu_w_abl32.so $P/libhstu_w_abl96.so $P/libhstu_w_abl256.so $P/libhstu_w_abl3.so $P/libhstu_w_abl8.so 2>&1 | tail -30
} > gpurun_out/r4/call02.txt 2>&1
tail -120 gpurun_out/r4/call02.txt
