#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r3/gputests37.txt
