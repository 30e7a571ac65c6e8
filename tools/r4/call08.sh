#!/bin/bash
# round 4, GPU call 8: the whole GPU suite (with the new fuzz slice), the reference's Triton kernel on this box, default bench
mkdir -p gpurun_out/r4
{
echo "== pytest -m gpu (all)"
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -25
echo "== bench default"
timeout 600 python bench.py > gpurun_out/r4/bench_call08_default.json 2> gpurun_out/r4/bench_call08_default.err; tail -c 3000 gpurun_out/r4/bench_call08_default.json
echo "== reference Triton comparator"
timeout 1500 python tools/ref_triton_compare.py 2>&1 | grep -v "amdgpu.ids" | tail -80
} > gpurun_out/r4/call08.txt 2>&1
tail -110 gpurun_out/r4/call08.txt
