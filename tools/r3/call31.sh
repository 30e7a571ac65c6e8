#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab_bwd.py --fwd tests/probe/libhstu_flip0.so tests/probe/libhstu_flip1.so > gpurun_out/r3/ab31.txt 2>&1
python tools/ab_bwd.py --fwd --workload M-jag tests/probe/libhstu_flip0.so tests/probe/libhstu_flip1.so >> gpurun_out/r3/ab31.txt 2>&1
python tools/ab_bwd.py --fwd --head-dim 64 tests/probe/libhstu_flip0.so tests/probe/libhstu_flip1.so >> gpurun_out/r3/ab31.txt 2>&1
python tools/ab_bwd.py --fwd --max-seq-len 256 tests/probe/libhstu_flip0.so tests/probe/libhstu_flip1.so >> gpurun_out/r3/ab31.txt 2>&1
cat gpurun_out/r3/ab31.txt | grep -v amdgpu.ids
