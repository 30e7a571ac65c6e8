#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab_norm.py generative_recommenders_amd/libhstu_hip.so tests/probe/libhstu_mc1.so > gpurun_out/r3/ab28.txt 2>&1
cat gpurun_out/r3/ab28.txt | tail -20
