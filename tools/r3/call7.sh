#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py --fwd $P/libhstu_base.so $P/libhstu_g0.so $P/libhstu_g1.so > $OUT/ab7.txt 2>&1
timeout 300 python tools/ab_bwd.py --fwd --workload M-jag $P/libhstu_base.so $P/libhstu_g0.so $P/libhstu_g1.so > $OUT/ab7_jag.txt 2>&1
cat $OUT/ab7.txt $OUT/ab7_jag.txt
