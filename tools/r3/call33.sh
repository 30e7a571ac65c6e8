#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_compute_gpu.py tests/test_dropout_gpu.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -15
python bench.py --no-extra --no-cpu --steps 10 --warmup 3 > gpurun_out/r3/bench33.json 2> gpurun_out/r3/bench33.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench33.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
l=d.get('layer'); print(l['ms_per_step'], l['dropout_off'], l['no_recompute'])
PY
python tools/ab_norm.py generative_recommenders_amd/libhstu_hip.so 2>&1 | grep -v amdgpu
