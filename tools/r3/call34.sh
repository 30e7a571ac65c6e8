#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_compute_gpu.py -m gpu -q 2>&1 | tail -8
python tools/prof_layer.py > gpurun_out/r3/prof_layer34.txt 2>&1
grep -A45 "device kernels:" gpurun_out/r3/prof_layer34.txt | cut -c1-200
