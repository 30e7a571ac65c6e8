#!/bin/bash
mkdir -p gpurun_out/r4
timeout 300 tools/ubench/gemm_stream_bench 2>&1 | tee gpurun_out/r4/gemm_stream_bench.txt
