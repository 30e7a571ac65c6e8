#!/bin/bash
OUT=gpurun_out/r05_call6
mkdir -p $OUT
{
L="tests/probe/libhstu_fold_base.so generative_recommenders_amd/libhstu_hip.so tests/probe/libhstu_fold_zero.so tests/probe/libhstu_fold_fma.so"
echo "== M-full"; timeout 300 python tools/ab_bwd.py --reps 7 $L 2>&1 | tail -5
echo "== M-jag"; timeout 300 python tools/ab_bwd.py --workload M-jag --reps 5 $L 2>&1 | tail -5
echo "== 1024 users"; timeout 300 python tools/ab_bwd.py --users 1024 --reps 9 --launches 30 $L 2>&1 | tail -5
echo "== d64"; timeout 300 python tools/ab_bwd.py --head-dim 64 --reps 5 $L 2>&1 | tail -5
echo "== tests"
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_metric_shapes_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x 2>&1 | tail -4
} > $OUT/log.txt 2>&1
cat $OUT/log.txt | cut -c1-250
