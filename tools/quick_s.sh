#!/bin/bash
# tools/quick_s.sh NAME "-DFLAG ..." [TU]: device assembly only of one translation unit -> csrc/build_var/NAME.s + register summary (CPU only)
cd "$(dirname "$0")/../generative_recommenders_amd/csrc"
mkdir -p build_var
TU=${3:-attn_fold_bf16}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -I. -I../../include --cuda-device-only -S $TU.hip -o build_var/$1.s 2>&1 | grep -E "error|warning: v" | head
python3 ../../tools/isa_census.py build_var/$1.s ${4:-kernel} | grep -A2 "^_Z" | grep -v "^--"
