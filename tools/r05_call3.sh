#!/bin/bash
OUT=gpurun_out/r05_call3
mkdir -p $OUT
{
echo "== A/B fold: counted vmcnt (default lib) vs vmcnt(0) (fold_vm0)"
timeout 300 python tools/ab_bwd.py --reps 7 generative_recommenders_amd/libhstu_hip.so tests/probe/libhstu_fold_vm0.so 2>&1 | tail -8
timeout 300 python tools/ab_bwd.py --workload M-jag --reps 5 generative_recommenders_amd/libhstu_hip.so tests/probe/libhstu_fold_vm0.so 2>&1 | tail -5
timeout 300 python tools/ab_bwd.py --users 1024 --reps 9 --launches 30 generative_recommenders_amd/libhstu_hip.so tests/probe/libhstu_fold_vm0.so 2>&1 | tail -5
echo "== attention tests (fold)"
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_metric_shapes_gpu.py -q -m gpu -x 2>&1 | tail -4
echo "== glue tests"
timeout 600 python -m pytest tests/test_glue_gpu.py -q -m gpu 2>&1 | tail -8
echo "== projections / layer"
timeout 600 python bench.py --no-cpu --no-extra --steps 10 --warmup 5 --parity-users 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); L=d['layer']; print('layer', L.get('ms_per_step'), L.get('error'), {k: L[k]['ms_per_step'] for k in ('two_node_layers','dropout_off','no_recompute') if k in L}); print(L['projections'].get('bias_grad'))"
HSTU_DBETA_STREAM=0 timeout 600 python bench.py --no-cpu --no-extra --steps 10 --warmup 5 --parity-users 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); L=d['layer']; print('layer dbeta in-stream', L.get('ms_per_step'), L.get('error'))"
} > $OUT/log.txt 2>&1
tail -60 $OUT/log.txt | cut -c1-500
