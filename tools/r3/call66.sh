#!/bin/bash
for lib in generative_recommenders_amd/libhstu_hip.so tests/probe/libhstu_posnext.so; do
echo "== $lib"
HSTU_HIP_LIBRARY=$PWD/$lib timeout 600 python tools/bench_ops.py 8192 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
for k,v in o['kernels'].items():
    if 'position' in k or 'add_timestamp' in k: print(f\"{k:60s} {v['us']:9.1f} us  {v['frac_of_hbm_peak']}\")"
done
HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_posnext.so timeout 600 python -m pytest tests/test_position_gpu.py -m gpu -q 2>&1 | tail -2
