#!/bin/bash
# round-3 GPU call 1: register-staged Q/dO tiles in the folded backward (A/B), cycle traces of both kernels
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py $P/libhstu_regs0.so $P/libhstu_regs1.so > $OUT/ab1.txt 2>&1
timeout 300 python tools/ab_bwd.py --workload M-jag $P/libhstu_regs0.so $P/libhstu_regs1.so > $OUT/ab1_jag.txt 2>&1
timeout 300 python tools/trace_run.py > $OUT/trace_bwd_regs1.txt 2>&1
timeout 300 python tools/trace_fwd.py > $OUT/trace_fwd.txt 2>&1
cat $OUT/ab1.txt $OUT/ab1_jag.txt; tail -70 $OUT/trace_fwd.txt
