#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab_norm.py generative_recommenders_amd/libhstu_hip.so tests/probe/libhstu_mc1.so tests/probe/libhstu_pre0.so tests/probe/libhstu_pre1.so > gpurun_out/r3/ab29.txt 2>&1
cat gpurun_out/r3/ab29.txt | tail -30
