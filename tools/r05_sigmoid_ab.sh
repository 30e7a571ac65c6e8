#!/bin/bash
# (record) row kernels: sigmoid through hardware exp2 / rcp (default) against expf + IEEE divide (tests/probe/libhstu_slowsig.so)
OUT=gpurun_out/r05_sigmoid
mkdir -p $OUT
{
timeout 900 python -m pytest tests/test_compute_gpu.py tests/test_swish_layer_norm_gpu.py tests/test_ln_linear_gpu.py tests/test_glue_gpu.py tests/test_fuzz_gpu.py tests/test_dropout_gpu.py tests/test_metric_shapes_gpu.py tests/test_research_gpu.py -q -m gpu -x 2>&1 | tail -4
for rep in 1 2; do
for lib in "" $PWD/tests/probe/libhstu_slowsig.so; do
echo "== lib=${lib:-default}"
HSTU_HIP_LIBRARY=$lib timeout 300 python bench.py --no-cpu --no-extra --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); L=d['layer']
print('layer ms', round(L['ms_per_step'],3), 'two-node', round(L['two_node_layers']['ms_per_step'],3), 'dropout_off', round(L['dropout_off']['ms_per_step'],3), 'no_recompute', round(L['no_recompute']['ms_per_step'],3))"
done; done
for lib in "" $PWD/tests/probe/libhstu_slowsig.so; do
HSTU_HIP_LIBRARY=$lib timeout 300 python tools/bench_ops.py 8192 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['kernels']
print({k: (v['us'], v['GBps']) for k,v in d.items() if 'norm' in k or 'silu' in k})"
done
} > $OUT/log.txt 2>&1
tail -30 $OUT/log.txt | cut -c1-600
