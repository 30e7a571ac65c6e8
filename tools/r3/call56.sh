#!/bin/bash
timeout 1200 python -m pytest tests/test_research_gpu.py tests/test_solo_gpu.py tests/test_configs_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python tools/bench_research.py books --kernels 2>&1 | grep -v amdgpu | grep "solo\|_ms"
timeout 300 python bench.py --workload C3 --steps 50 --warmup 10 --no-layer --no-cpu --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3', round(d['value']), 'fwd', round(d['roofline_fwd']['avg_launch_ms']*1e3,1), 'bwd', round(d['roofline']['avg_launch_ms']*1e3,1))"
