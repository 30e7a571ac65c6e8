#!/bin/bash
# round 4, GPU call 11: new tests (precise forward, deterministic dq), traffic passes for the extra workloads, precise-forward speed
mkdir -p gpurun_out/r4
{
echo "== pytest (changed areas)"
timeout 900 python -m pytest tests/test_precise_gpu.py tests/test_attention_gpu.py tests/test_compute_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -6
cat gpurun_out/precise_forward_errors.json
echo "== precise forward speed"
for pz in 0 1; do HSTU_ATTN_PRECISE=$pz timeout 300 python bench.py --steps 20 --warmup 5 --no-layer --no-cpu --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('precise=$pz fwd ms', round(d['roofline_fwd']['avg_launch_ms'],4), d['roofline_fwd'].get('kernel'), 'bwd ms', round(d['roofline']['avg_launch_ms'],4))"; done
echo "== traffic passes"
bash tools/prof_traffic.sh r04_traffic_Mjag --workload M-jag > /dev/null 2>&1; grep -A12 "PMC counters" gpurun_out/prof_r04_traffic_Mjag/summary.md | head -40
bash tools/prof_traffic.sh r04_traffic_d64 --head-dim 64 > /dev/null 2>&1; grep -A12 "PMC counters" gpurun_out/prof_r04_traffic_d64/summary.md | head -40
bash tools/prof_traffic.sh r04_traffic_M1024 --users-per-gpu 1024 > /dev/null 2>&1; grep -A12 "PMC counters" gpurun_out/prof_r04_traffic_M1024/summary.md | head -40
} > gpurun_out/r4/call11.txt 2>&1
tail -120 gpurun_out/r4/call11.txt | cut -c1-250
