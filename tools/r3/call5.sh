#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py $P/libhstu_base.so $P/libhstu_sip0.so $P/libhstu_sip1.so > $OUT/ab5.txt 2>&1
timeout 300 python tools/ab_bwd.py --workload M-jag $P/libhstu_base.so $P/libhstu_sip0.so $P/libhstu_sip1.so > $OUT/ab5_jag.txt 2>&1
cat $OUT/ab5.txt $OUT/ab5_jag.txt
