#!/bin/bash
# (record) hstu_linear_k512: parity tests, then the projection table and the layer step with and without it
OUT=gpurun_out/r05_k512
mkdir -p $OUT
{
timeout 600 python -m pytest tests/test_ln_linear_gpu.py tests/test_metric_shapes_gpu.py -q -m gpu -x 2>&1 | tail -5
for e in 1 0 1 0; do
echo "== HSTU_OUT_DGRAD_KERNEL=$e"
HSTU_OUT_DGRAD_KERNEL=$e timeout 300 python bench.py --no-cpu --no-extra --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); L=d['layer']; p=L['projections']
print('layer ms', round(L['ms_per_step'],3), 'two-node', round(L['two_node_layers']['ms_per_step'],3), 'out_dgrad', p['out_dgrad']['us'], 'k512', p.get('out_dgrad_k512',{}).get('us'), 'uvqk_fused', p['uvqk_fwd_fused']['us'], 'value', round(d['value']))"
done
} > $OUT/log.txt 2>&1
tail -30 $OUT/log.txt | cut -c1-300
