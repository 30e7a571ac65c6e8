#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
T="tests/test_configs_gpu.py::test_research_configs_rel_bias_attention"
for lib in "" tests/probe/libhstu_bf_fast0.so tests/probe/libhstu_bf_asm0.so tests/probe/libhstu_bf_both0.so; do
  echo "=== $lib"; HSTU_HIP_LIBRARY=$lib timeout 600 python -m pytest "$T" -x -q 2>&1 | tail -3
done
