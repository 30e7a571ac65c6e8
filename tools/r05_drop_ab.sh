#!/bin/bash
# (record) fused dropout generator: one multiply-xorshift round per element pair (default) against two (tests/probe/libhstu_drop2.so, timing only)
OUT=gpurun_out/r05_drop
mkdir -p $OUT
{
timeout 600 python -m pytest tests/test_dropout_gpu.py tests/test_compute_gpu.py tests/test_fuzz_gpu.py tests/test_ln_linear_gpu.py tests/test_glue_gpu.py tests/test_swish_layer_norm_gpu.py tests/test_metric_shapes_gpu.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | tail -4
for rep in 1 2; do
for lib in "" $PWD/tests/probe/libhstu_drop2.so; do
HSTU_HIP_LIBRARY=$lib timeout 300 python bench.py --no-cpu --no-extra --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); L=d['layer']
print('lib=${lib:-default}', 'layer ms', round(L['ms_per_step'],3), 'two-node', round(L['two_node_layers']['ms_per_step'],3), 'dropout_off', round(L['dropout_off']['ms_per_step'],3))"
done; done
} > $OUT/log.txt 2>&1
cat $OUT/log.txt | tail -12 | cut -c1-300
