#!/bin/bash
OUT=gpurun_out/r05_call4
mkdir -p $OUT
{
echo "== glue tests"
timeout 600 python -m pytest tests/test_glue_gpu.py -q -m gpu -x 2>&1 | tail -40
echo "== metric shapes"
timeout 600 python -m pytest tests/test_metric_shapes_gpu.py -q -m gpu 2>&1 | tail -4
} > $OUT/log.txt 2>&1
timeout 1200 python tools/ref_triton_compare.py --workloads C2-len --max-seq-len 211 --head-dim 64 --iters 20 2>&1 | grep -v "Warning\|warn" | awk '!seen[$0]++' > $OUT/triton_c2.txt
tail -60 $OUT/log.txt | cut -c1-300; tail -12 $OUT/triton_c2.txt | cut -c1-400
