#!/bin/bash
# round 4, GPU call 16: final validation -- whole GPU suite, smoke, default bench, rocprofv3 stats + PMC of the final code
mkdir -p gpurun_out/r4
{
echo "== pytest -m gpu (all)"
timeout 1800 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -14
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== bench default"
timeout 900 python bench.py > gpurun_out/r4/bench_final_default.json 2> gpurun_out/r4/bench_final_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_final_default.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'ms/step', d['ms_per_step'], 'fwd', d['roofline_fwd']['avg_launch_ms'], d['roofline_fwd']['frac'], 'bwd', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'both', d.get('roofline_fwd_bwd',{}).get('frac'))
for k,v in d['extra_workloads'].items(): print(' ', k, {x: v.get(x) for x in ('fwd_ms','bwd_ms','frac_fwd','frac_bwd','frac_fwd_bwd','traffic')})
print('layer', {k: d['layer'].get(k) for k in ('ms_per_step','user_seqs_per_s')} if isinstance(d.get('layer'), dict) else d.get('layer'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
echo "== rocprofv3"
bash tools/prof_pmc.sh r04_final > /dev/null 2>&1; head -24 gpurun_out/prof_r04_final/summary.md
} > gpurun_out/r4/call16.txt 2>&1
tail -60 gpurun_out/r4/call16.txt | cut -c1-330
