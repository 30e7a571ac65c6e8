#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py --fwd $P/libhstu_base.so $P/libhstu_ns3.so $P/libhstu_ns4.so $P/libhstu_ns5.so > $OUT/ab18.txt 2>&1; cat $OUT/ab18.txt
timeout 300 python tools/ab_bwd.py --fwd --workload M-jag $P/libhstu_base.so $P/libhstu_ns5.so 2>&1 | tail -2
timeout 300 python tools/ab_bwd.py --fwd --head-dim 64 $P/libhstu_base.so $P/libhstu_ns5.so 2>&1 | tail -2
