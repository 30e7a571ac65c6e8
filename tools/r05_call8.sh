#!/bin/bash
OUT=gpurun_out/r05_call8
mkdir -p $OUT
{
echo "== research / configs tests"
timeout 900 python -m pytest tests/test_research_gpu.py tests/test_configs_gpu.py tests/test_fuzz_gpu.py -q -m gpu 2>&1 | tail -4
echo "== bias sweep slice"
timeout 600 python tools/fuzz_attention.py --cases 80 --seed 77 --bias 2>&1 | tail -3
echo "== C2"
for i in 1 2; do timeout 300 python bench.py --workload C2 --steps 20 --warmup 5 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C2', round(d['value']), 'fwd', round(d['roofline_fwd']['avg_launch_ms'],4), 'bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), d['roofline']['kernel'])"; done
} > $OUT/log.txt 2>&1
cat $OUT/log.txt | cut -c1-250
