#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab_bwd.py --fwd tests/probe/libhstu_ta0.so tests/probe/libhstu_ta1.so > gpurun_out/r3/ab68.txt 2>&1
python tools/ab_bwd.py --fwd --workload M-jag tests/probe/libhstu_ta0.so tests/probe/libhstu_ta1.so >> gpurun_out/r3/ab68.txt 2>&1
python tools/ab_bwd.py --fwd --head-dim 64 tests/probe/libhstu_ta0.so tests/probe/libhstu_ta1.so >> gpurun_out/r3/ab68.txt 2>&1
python tools/ab_bwd.py --fwd --max-seq-len 256 tests/probe/libhstu_ta0.so tests/probe/libhstu_ta1.so >> gpurun_out/r3/ab68.txt 2>&1
python tools/ab_bwd.py --fwd --max-seq-len 160 tests/probe/libhstu_ta0.so tests/probe/libhstu_ta1.so >> gpurun_out/r3/ab68.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3/ab68.txt
