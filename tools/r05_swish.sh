#!/bin/bash
# (record) swish_layer_norm: parity tests, then the row-kernel roofline table at 1024 and 8192 users
OUT=gpurun_out/r05_swish
mkdir -p $OUT
{
timeout 600 python -m pytest tests/test_swish_layer_norm_gpu.py tests/test_compute_gpu.py -q -m gpu 2>&1 | tail -8
timeout 300 python tools/bench_ops.py 8192 > $OUT/bench_ops_8192.json 2>$OUT/bench_ops.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_swish/bench_ops_8192.json'))
for k,v in d.items():
    if 'norm' in k: print(k, v)
PY
} > $OUT/log.txt 2>&1
tail -40 $OUT/log.txt | cut -c1-400
