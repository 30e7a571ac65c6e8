#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : libhstu_hip.so with attn_fold_bf16.hip recompiled under extra flags
# -> tests/probe/libhstu_NAME.so (A/B measurements with tools/ab.sh)
set -e
cd "$(dirname "$0")/../generative_recommenders_amd/csrc"
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -I. -I../../include -c attn_fold_bf16.hip -o build_var/$1.o
objs=$(ls build/*.o | grep -v attn_fold_bf16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_var/$1.o -o ../../tests/probe/libhstu_$1.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -I. -I../../include --cuda-device-only -S attn_fold_bf16.hip -o build_var/$1.s 2>/dev/null
echo "$1: $(grep -E 'vgpr_count|vgpr_spill' build_var/$1.s | head -2 | tr -s ' ' | tr '\n' ' ')"
