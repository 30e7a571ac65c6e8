#!/bin/bash
mkdir -p gpurun_out/r4
P=$PWD/tests/probe; B=generative_recommenders_amd/libhstu_hip.so
{
timeout 600 python tools/ab_norm.py $B $P/libhstu_norm_nt1.so $P/libhstu_norm_nt2.so $P/libhstu_norm_nt3.so 2>&1 | grep -v amdgpu | tail -30
for v in "" norm_nt1 norm_nt2 norm_nt3; do
  if [ -z "$v" ]; then unset HSTU_HIP_LIBRARY; else export HSTU_HIP_LIBRARY=$P/libhstu_$v.so; fi
  timeout 300 python bench.py --no-extra --no-cpu --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); L=d['layer']; print('layer', '$v', L.get('ms_per_step'), 'no_recompute', L.get('no_recompute',{}).get('ms_per_step'))"
done
} | tee gpurun_out/r4/norm_nt.txt
