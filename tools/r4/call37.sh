#!/bin/bash
# round 4, final validation after the fused LayerNorm + projection kernel: whole GPU suite, smoke, default bench,
# rocprofv3 of the layer section (per-kernel table with MFMA pipe busy) and of the projection micro-bench
mkdir -p gpurun_out/r4
{
echo "== pytest -m gpu (all)"
timeout 1800 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -12
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== bench default"
timeout 900 python bench.py > gpurun_out/r4/bench_final2_default.json 2> gpurun_out/r4/bench_final2_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_final2_default.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'ms/step', d['ms_per_step'], 'fwd', d['roofline_fwd']['avg_launch_ms'], d['roofline_fwd']['frac'], 'bwd', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'both', d.get('roofline_fwd_bwd',{}).get('frac'))
for k,v in d['extra_workloads'].items(): print(' ', k, {x: v.get(x) for x in ('fwd_ms','bwd_ms','frac_fwd','frac_bwd','frac_fwd_bwd')})
L=d.get('layer'); print('layer', {k: L.get(k) for k in ('ms_per_step','error')} , {k: L[k]['ms_per_step'] for k in ('two_node_layers','dropout_off','no_recompute') if k in L}, L.get('projections'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
echo "== rocprofv3: layer section"
bash tools/prof_layer_pmc.sh r04 > /dev/null 2>&1; head -30 gpurun_out/prof_layer_r04/summary.md
echo "== rocprofv3: projection micro-bench"
bash tools/prof_cmd.sh r04_lnl_final python tools/bench_ln_linear.py --iters 10 > /dev/null 2>&1; head -22 gpurun_out/prof_r04_lnl_final/summary.md
} > gpurun_out/r4/call37.txt 2>&1
tail -90 gpurun_out/r4/call37.txt | cut -c1-400
