#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
timeout 2000 python -m pytest tests -m gpu -x -q > $OUT/gputest2.txt 2>&1; tail -15 $OUT/gputest2.txt
timeout 900 python bench.py > $OUT/bench2.json 2> $OUT/bench2.err; tail -c 600 $OUT/bench2.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline_fwd']['avg_launch_ms'], d['roofline']['avg_launch_ms'], d['roofline_fwd_bwd']['frac'])
print(json.dumps(d.get('extra_workloads'), indent=1)[:3000])
print(json.dumps({k:v for k,v in d.get('layer',{}).items() if k!='projections'}, indent=1)[:1500])
PY
