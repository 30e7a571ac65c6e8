#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py --fwd --reps 7 $P/libhstu_base.so $P/libhstu_ke0.so $P/libhstu_ke1.so > $OUT/ab26.txt 2>&1; cat $OUT/ab26.txt
timeout 300 python tools/ab_bwd.py --fwd --workload M-jag $P/libhstu_base.so $P/libhstu_ke0.so $P/libhstu_ke1.so 2>&1 | tail -3
timeout 300 python tools/ab_bwd.py --fwd --head-dim 64 $P/libhstu_base.so $P/libhstu_ke0.so $P/libhstu_ke1.so 2>&1 | tail -3
