#!/bin/bash
mkdir -p gpurun_out/r4
P=tests/probe; B=generative_recommenders_amd/libhstu_hip.so
{
echo "== backward, M-full / M-jag: tile requests with the nt hint"
timeout 600 python tools/ab_bwd.py $B $P/libhstu_fold_dnt.so 2>&1 | grep -v amdgpu | tail -3
timeout 600 python tools/ab_bwd.py --workload M-jag $B $P/libhstu_fold_dnt.so 2>&1 | grep -v amdgpu | tail -3
echo "== forward"
timeout 600 python tools/ab_bwd.py --fwd $B $P/libhstu_fwd_dnt.so 2>&1 | grep -v amdgpu | tail -3
timeout 600 python tools/ab_bwd.py --fwd --workload M-jag $B $P/libhstu_fwd_dnt.so 2>&1 | grep -v amdgpu | tail -3
} | tee gpurun_out/r4/dma_nt.txt
