#!/bin/bash
# round 4, GPU call 1: first run of the wide backward (hstu_attn_bwd_wide.cuh): parity, then A/B against the folded kernel
mkdir -p gpurun_out/r4
{
echo "== pytest attention + metric shapes (wide on)"
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_metric_shapes_gpu.py -x -q -m gpu 2>&1 | tail -15
echo "== fuzz (wide on)"
timeout 600 python tools/fuzz_attention.py --cases 120 --seed 41 2>&1 | tail -12
timeout 600 python tools/fuzz_attention.py --big --cases 8 --seed 42 2>&1 | tail -6
echo "== bench A/B"
for rep in 1 2; do for w in 0 1; do
HSTU_BWD_WIDE=$w timeout 300 python bench.py --steps 20 --warmup 5 --no-layer --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('wide=$w value', round(d['value']), 'fwd ms', round(d['roofline_fwd']['avg_launch_ms'],3), 'bwd ms', round(d['roofline']['avg_launch_ms'],3), d['roofline']['kernel'], 'M-jag', (d.get('extra_workloads',{}).get('M-jag',{}) or {}).get('frac_bwd'))"
done; done
} > gpurun_out/r4/call01.txt 2>&1
tail -40 gpurun_out/r4/call01.txt
