#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py $P/libhstu_base.so $P/libhstu_dc0.so $P/libhstu_dc10.so $P/libhstu_dc1.so > $OUT/ab20.txt 2>&1; cat $OUT/ab20.txt
timeout 300 python tools/ab_bwd.py --workload M-jag $P/libhstu_dc0.so $P/libhstu_dc10.so $P/libhstu_dc1.so 2>&1 | tail -3
timeout 300 python tools/ab_bwd.py --fwd --reps 9 $P/libhstu_es0.so $P/libhstu_es1.so > $OUT/ab20f.txt 2>&1; cat $OUT/ab20f.txt
