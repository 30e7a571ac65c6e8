#!/bin/bash
mkdir -p gpurun_out/r3
for f in 0 1 0 1; do
HSTU_UVQK_LINEAR=$f python bench.py --no-extra --no-cpu --steps 5 --warmup 2 --layer-steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['layer']; print('linear=$f', round(l['ms_per_step'],3), round(l['dropout_off']['ms_per_step'],3), round(l['no_recompute']['ms_per_step'],3), l['projections']['uvqk_fwd'])"
done
python -m pytest tests/test_compute_gpu.py -m gpu -q 2>&1 | tail -3
