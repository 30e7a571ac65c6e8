#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_ln_linear_gpu.py tests/test_compute_gpu.py tests/test_research_gpu.py -x -q 2>&1 | tail -6
for s in 48 96 192; do timeout 300 python bench.py --workload M-full --users 1024 --steps $s --warmup 20 --no-layer --no-cpu --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps', d['steps'], 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', d['roofline']['avg_launch_ms'], 'both', d['roofline_fwd_bwd']['frac'])"; done
