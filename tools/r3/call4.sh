#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py $P/libhstu_base.so $P/libhstu_regs0.so $P/libhstu_regs1.so $P/libhstu_regs1late.so > $OUT/ab4.txt 2>&1
timeout 300 python tools/ab_bwd.py --workload M-jag $P/libhstu_base.so $P/libhstu_regs1.so $P/libhstu_regs1late.so > $OUT/ab4_jag.txt 2>&1
cat $OUT/ab4.txt $OUT/ab4_jag.txt
