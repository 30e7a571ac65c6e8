#!/bin/bash
# A/B of library variants on the metric shape: tools/ab.sh lib1.so lib2.so ...  (prints fwd / bwd ms per launch)
for rep in 1 2; do
for lib in "$@"; do
  HSTU_HIP_LIBRARY=$lib python bench.py --steps 20 --warmup 5 --no-layer --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib'.split('/')[-1], 'value', round(d['value']), 'fwd ms', round(d['roofline_fwd']['avg_launch_ms'],3), 'bwd ms', round(d['roofline']['avg_launch_ms'],3), d['roofline']['kernel'])"
done
done
