#!/bin/bash
for lib in generative_recommenders_amd/libhstu_hip.so tests/probe/libhstu_sa2.so tests/probe/libhstu_sa1.so tests/probe/libhstu_sa3.so tests/probe/libhstu_sa7.so; do
echo "== $lib"
HSTU_HIP_LIBRARY=$PWD/$lib timeout 300 python tools/bench_research.py books --kernels 2>&1 | grep "fwd_solo_bias"
done
