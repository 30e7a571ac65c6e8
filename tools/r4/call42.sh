#!/bin/bash
mkdir -p gpurun_out/r4
P=tests/probe
timeout 600 python tools/ab_bwd.py $P/libhstu_fold_nt0.so $P/libhstu_fold_nt1.so $P/libhstu_fold_nt2.so $P/libhstu_fold_nt3.so 2>&1 | tail -8 | tee gpurun_out/r4/fold_nt_stores.txt
timeout 600 python tools/ab_bwd.py --workload M-jag $P/libhstu_fold_nt0.so $P/libhstu_fold_nt1.so $P/libhstu_fold_nt2.so $P/libhstu_fold_nt3.so 2>&1 | tail -6 | tee -a gpurun_out/r4/fold_nt_stores.txt
