#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py $P/libhstu_regs0.so $P/libhstu_regs1.so $P/libhstu_regs1full.so $P/libhstu_regs1late.so > $OUT/ab2.txt 2>&1
timeout 300 python tools/ab_bwd.py --workload M-jag $P/libhstu_regs0.so $P/libhstu_regs1.so $P/libhstu_regs1late.so > $OUT/ab2_jag.txt 2>&1
cat $OUT/ab2.txt $OUT/ab2_jag.txt
