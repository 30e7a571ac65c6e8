#!/bin/bash
OUT=gpurun_out/r05_call5
mkdir -p $OUT
{
echo "== glue tests"
timeout 600 python -m pytest tests/test_glue_gpu.py -q -m gpu 2>&1 | tail -30
echo "== compute / ln_linear / metric shapes / data-parallel-independent tests"
timeout 900 python -m pytest tests/test_compute_gpu.py tests/test_ln_linear_gpu.py tests/test_metric_shapes_gpu.py -q -m gpu 2>&1 | tail -8
echo "== projections / layer"
timeout 600 python bench.py --no-cpu --no-extra --steps 10 --warmup 5 --parity-users 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); L=d['layer']; print('layer', L.get('ms_per_step'), L.get('error'), {k: L[k]['ms_per_step'] for k in ('two_node_layers','dropout_off','no_recompute') if k in L}); print(L['projections'].get('bias_grad'))"
} > $OUT/log.txt 2>&1
tail -60 $OUT/log.txt | cut -c1-300
