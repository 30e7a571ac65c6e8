#!/bin/bash
mkdir -p gpurun_out/r3
python tools/prof_layer.py > gpurun_out/r3/prof_layer27.txt 2>&1
tail -80 gpurun_out/r3/prof_layer27.txt | cut -c1-230
