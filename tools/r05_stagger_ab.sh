#!/bin/bash
# (record) A/B of FOLD_STAGGER (persistent workgroups of the folded backward start out of phase) against the default build
OUT=gpurun_out/r05_stagger
mkdir -p $OUT
P=tests/probe
{
echo "== backward M-full 8192 users"
timeout 300 python tools/ab_bwd.py --workload M-full --reps 7 $P/libhstu_base0.so $P/libhstu_st1.so $P/libhstu_st2.so $P/libhstu_st4.so
echo "== backward M-full 1024 users"
timeout 300 python tools/ab_bwd.py --workload M-full --users 1024 --reps 9 --launches 20 $P/libhstu_base0.so $P/libhstu_st1.so $P/libhstu_st2.so $P/libhstu_st4.so
echo "== backward M-jag"
timeout 300 python tools/ab_bwd.py --workload M-jag --reps 7 $P/libhstu_base0.so $P/libhstu_st1.so $P/libhstu_st2.so $P/libhstu_st4.so
} > $OUT/log.txt 2>&1
tail -30 $OUT/log.txt | cut -c1-300
