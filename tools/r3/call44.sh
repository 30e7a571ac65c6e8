#!/bin/bash
for wl in M-jag M-targets; do
python bench.py --workload $wl --users-per-gpu 1024 --no-layer --no-cpu --no-extra --steps 50 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl 1024 users: fwd', round(d['roofline_fwd']['avg_launch_ms']*1e3,1), 'us bwd', round(d['roofline']['avg_launch_ms']*1e3,1), 'us', d['roofline']['kernel'])"
done
for s in 0 1; do
python bench.py --workload M-targets --users-per-gpu 1024 --no-layer --no-cpu --no-extra --steps 50 --warmup 10 --sort-by-length $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('M-targets sort=$s: fwd', round(d['roofline_fwd']['avg_launch_ms']*1e3,1), 'us bwd', round(d['roofline']['avg_launch_ms']*1e3,1))"
done
