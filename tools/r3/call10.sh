#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py --fwd $P/libhstu_base.so $P/libhstu_k0.so $P/libhstu_k2.so $P/libhstu_k3w.so $P/libhstu_k4w.so $P/libhstu_k3v2w.so $P/libhstu_k3.so > $OUT/ab10.txt 2>&1
cat $OUT/ab10.txt
