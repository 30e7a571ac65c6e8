#!/bin/bash
mkdir -p gpurun_out/r4
{
for lib in generative_recommenders_amd/libhstu_hip.so tests/probe/libhstu_w2_nostream.so tests/probe/libhstu_w2_p1.so; do
echo "=== $lib"; HSTU_HIP_LIBRARY=$PWD/$lib timeout 300 python tools/r4/dbg_wide.py 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r4/call05.txt 2>&1
cat gpurun_out/r4/call05.txt
