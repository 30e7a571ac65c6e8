#!/bin/bash
mkdir -p gpurun_out/r3
for wl in C3-bias C3 C2; do
timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --no-layer --no-cpu --no-extra > gpurun_out/r3/bench_e_$wl.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r3/bench_e_$wl.json').read().strip().splitlines()[-1])
print('$wl', round(d['value']), 'fwd', round(d['roofline_fwd']['avg_launch_ms'],3), 'bwd', round(d['roofline']['avg_launch_ms'],3), round(d['roofline_fwd_bwd']['frac'],3), d['roofline']['kernel'], d['roofline_fwd']['kernel'])
PY
done
