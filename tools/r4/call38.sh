#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -q -m gpu --durations=3 2>&1 | tail -12 | tee gpurun_out/r4/call38.txt
