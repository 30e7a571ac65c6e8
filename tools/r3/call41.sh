#!/bin/bash
mkdir -p gpurun_out/r3
timeout 300 tools/gemm_exp/lt_probe 2>&1 | tee gpurun_out/r3/lt_probe41.txt
