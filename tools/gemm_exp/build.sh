#!/bin/bash
# tools/gemm_exp/build.sh : the hand-written projection GEMM experiment -> tools/gemm_exp/libgemm_exp.so
# (stand-alone: stubs.hip supplies the error plumbing the product keeps in capi.hip)
set -e
cd "$(dirname "$0")"
CSRC=../../generative_recommenders_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $* -I$CSRC -I../../include -c gemm_ops.hip -o gemm_ops.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -I$CSRC -I../../include -c stubs.hip -o stubs.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gemm_ops.o stubs.o -o libgemm_exp.so
rm -f gemm_ops.o stubs.o
echo built tools/gemm_exp/libgemm_exp.so
