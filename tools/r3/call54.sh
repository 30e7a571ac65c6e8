#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_research_gpu.py -m gpu -q -x -k "short_sequence or rel_bias" 2>&1 | tail -4
for w in 1 0; do
echo "== HSTU_SOLO_BIAS_WAVE=$w"
HSTU_SOLO_BIAS_WAVE=$w timeout 300 python tools/bench_research.py books --kernels 2>&1 | grep "solo_bias"
done
