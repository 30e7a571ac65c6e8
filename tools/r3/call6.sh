#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py --fwd $P/libhstu_base.so $P/libhstu_f000.so $P/libhstu_f100.so $P/libhstu_f010.so $P/libhstu_f001.so $P/libhstu_f111.so > $OUT/ab6.txt 2>&1
timeout 300 python tools/ab_bwd.py --fwd --workload M-jag $P/libhstu_base.so $P/libhstu_f000.so $P/libhstu_f100.so $P/libhstu_f111.so > $OUT/ab6_jag.txt 2>&1
cat $OUT/ab6.txt $OUT/ab6_jag.txt
