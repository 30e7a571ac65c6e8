#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
timeout 300 python tools/trace_run.py > $OUT/trace_bwd_regs1b.txt 2>&1
cat $OUT/trace_bwd_regs1b.txt
