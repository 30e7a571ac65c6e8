#!/bin/bash
mkdir -p gpurun_out/r3
timeout 600 python tools/prof_layer.py > gpurun_out/r3/prof_layer.txt 2>&1; tail -45 gpurun_out/r3/prof_layer.txt | cut -c1-200
