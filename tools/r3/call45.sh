#!/bin/bash
for s in 0 1; do
python bench.py --workload M-targets --users-per-gpu 1024 --no-layer --no-cpu --no-extra --steps 50 --warmup 10 --sort-by-length $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('M-targets 1024 sort=$s: fwd', round(d['roofline_fwd']['avg_launch_ms']*1e3,1), 'us bwd', round(d['roofline']['avg_launch_ms']*1e3,1))"
python bench.py --workload C3 --no-layer --no-cpu --no-extra --steps 50 --warmup 10 --sort-by-length $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 sort=$s: fwd', round(d['roofline_fwd']['avg_launch_ms']*1e3,1), 'us bwd', round(d['roofline']['avg_launch_ms']*1e3,1))"
python bench.py --workload C2 --no-layer --no-cpu --no-extra --steps 20 --warmup 5 --sort-by-length $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 sort=$s: fwd', round(d['roofline_fwd']['avg_launch_ms']*1e3,1), 'us bwd', round(d['roofline']['avg_launch_ms']*1e3,1))"
done
python -m pytest tests/test_attention_gpu.py -m gpu -q -k "sort or order" 2>&1 | tail -3
