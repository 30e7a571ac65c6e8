#!/bin/bash
mkdir -p gpurun_out/r05_call9
timeout 600 python tools/trace_layer_step.py > gpurun_out/r05_call9/trace.txt 2>&1
cat gpurun_out/r05_call9/trace.txt | cut -c1-200 | tail -70
