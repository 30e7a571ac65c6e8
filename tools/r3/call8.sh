#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py --fwd $P/libhstu_base.so $P/libhstu_g1.so $P/libhstu_a0.so $P/libhstu_a1.so $P/libhstu_a1u0.so $P/libhstu_a1e.so $P/libhstu_a1d0.so > $OUT/ab8.txt 2>&1
timeout 300 python tools/ab_bwd.py --fwd --workload M-jag $P/libhstu_base.so $P/libhstu_g1.so $P/libhstu_a1.so $P/libhstu_a1e.so > $OUT/ab8_jag.txt 2>&1
cat $OUT/ab8.txt $OUT/ab8_jag.txt
