#!/bin/bash
# round 4, GPU call 10: rocprofv3 kernel trace + PMC passes of the attention bench: folded backward (default) and wide (opt-in)
mkdir -p gpurun_out/r4
bash tools/prof_pmc.sh r04_fold > gpurun_out/r4/call10_fold.txt 2>&1
HSTU_BWD_WIDE=1 bash tools/prof_pmc.sh r04_wide > gpurun_out/r4/call10_wide.txt 2>&1
timeout 300 python bench.py --no-layer --no-cpu > gpurun_out/r4/bench_call10.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_call10.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', d['roofline']['avg_launch_ms'])
for k,v in d['extra_workloads'].items(): print(k, {x: v.get(x) for x in ('fwd_ms','bwd_ms','frac_fwd','frac_bwd','frac_fwd_bwd','steps')})
PY
tail -45 gpurun_out/r4/call10_fold.txt | cut -c1-220; tail -45 gpurun_out/r4/call10_wide.txt | cut -c1-220
