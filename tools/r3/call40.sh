#!/bin/bash
mkdir -p gpurun_out/r3
python tools/prof_layer.py > gpurun_out/r3/prof_layer40.txt 2>&1
grep "^{'users" gpurun_out/r3/prof_layer40.txt | cut -c1-200
grep -A40 "device kernels:" gpurun_out/r3/prof_layer40.txt | cut -c1-170
