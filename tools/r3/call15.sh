#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
timeout 2000 python -m pytest tests -m gpu -q > $OUT/gputest3.txt 2>&1; tail -8 $OUT/gputest3.txt
