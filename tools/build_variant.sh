#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." [TU] : libhstu_hip.so with one translation unit (default attn_fold_bf16)
# recompiled under extra flags
# -> tests/probe/libhstu_NAME.so (A/B measurements with tools/ab.sh)
set -e
cd "$(dirname "$0")/../generative_recommenders_amd/csrc"
mkdir -p build_var
TU=${3:-attn_fold_bf16}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -I. -I../../include -c $TU.hip -o build_var/$1.o
objs=$(ls build/*.o | grep -v $TU.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_var/$1.o -o ../../tests/probe/libhstu_$1.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -I. -I../../include --cuda-device-only -S $TU.hip -o build_var/$1.s 2>/dev/null
echo "$1: $(grep -E 'vgpr_count|vgpr_spill' build_var/$1.s | head -4 | tr -s ' ' | tr '\n' ' ')"
