#!/bin/bash
mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_research_gpu.py tests/test_configs_gpu.py tests/test_solo_gpu.py tests/test_attention_gpu.py -m gpu -q -x 2>&1 | tail -4
timeout 300 python tools/bench_research.py books --kernels 2>&1 | grep -v amdgpu | grep "solo_bias\|_ms" | tee gpurun_out/r3/research_books55.txt
timeout 300 python bench.py --workload C3-bias --steps 20 --warmup 5 --no-layer --no-cpu --no-extra > gpurun_out/r3/bench_f_C3-bias.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r3/bench_f_C3-bias.json').read().strip().splitlines()[-1])
print('C3-bias', round(d['value']), 'fwd', round(d['roofline_fwd']['avg_launch_ms'],3), 'bwd', round(d['roofline']['avg_launch_ms'],3), round(d['roofline_fwd_bwd']['frac'],3))"
