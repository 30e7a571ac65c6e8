#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
timeout 300 python tools/trace_fwd.py > $OUT/trace_fwd2.txt 2>&1
cat $OUT/trace_fwd2.txt
