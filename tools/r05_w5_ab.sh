#!/bin/bash
# (record) u * GroupNorm(attn) backward with SiLU on the fly: 5 waves per SIMD with 6 spilled registers (default) against 4 waves without (-DNORM_W5=0)
OUT=gpurun_out/r05_w5
mkdir -p $OUT
{
for rep in 1 2 3; do
for lib in "" $PWD/tests/probe/libhstu_w4.so; do
HSTU_HIP_LIBRARY=$lib timeout 300 python bench.py --no-cpu --no-extra --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); L=d['layer']
print('lib=${lib:-default}', 'layer ms', round(L['ms_per_step'],3), 'two-node', round(L['two_node_layers']['ms_per_step'],3), 'dropout_off', round(L['dropout_off']['ms_per_step'],3))"
done; done
} > $OUT/log.txt 2>&1
cat $OUT/log.txt | cut -c1-300
