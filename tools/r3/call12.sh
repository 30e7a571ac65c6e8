#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
P=tests/probe
timeout 300 python tools/ab_bwd.py $P/libhstu_base.so $P/libhstu_fd0.so $P/libhstu_fd1.so > $OUT/ab12.txt 2>&1
timeout 300 python tools/ab_bwd.py --workload M-jag $P/libhstu_base.so $P/libhstu_fd0.so $P/libhstu_fd1.so > $OUT/ab12_jag.txt 2>&1
timeout 300 python tools/ab_bwd.py --head-dim 64 $P/libhstu_base.so $P/libhstu_fd0.so $P/libhstu_fd1.so > $OUT/ab12_d64.txt 2>&1
cat $OUT/ab12.txt $OUT/ab12_jag.txt $OUT/ab12_d64.txt
