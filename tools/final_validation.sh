#!/bin/bash
# End-of-round validation on the GPU box (gpurun -- 'bash tools/final_validation.sh [tag]'): the whole GPU suite, smoke, the default
# bench line, rocprofv3 kernel trace + PMC of the attention bench, of the layer section (per-kernel table with MFMA pipe busy) and of the
# fused projection's micro-bench.  Everything lands under gpurun_out/; copy what is to be judged into profiles/.
TAG=${1:-final}
mkdir -p gpurun_out/${TAG}
{
echo "== pytest -m gpu (all)"
timeout 1800 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -12
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== bench default"
timeout 900 python bench.py > gpurun_out/${TAG}/bench_default.json 2> gpurun_out/${TAG}/bench_default.err; TAG=${TAG} python - <<'PY'
import json, os
d=json.loads(open('gpurun_out/' + os.environ['TAG'] + '/bench_default.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'ms/step', d['ms_per_step'], 'fwd', d['roofline_fwd']['avg_launch_ms'], d['roofline_fwd']['frac'], 'bwd', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'both', d.get('roofline_fwd_bwd',{}).get('frac'))
for k,v in d['extra_workloads'].items(): print(' ', k, {x: v.get(x) for x in ('fwd_ms','bwd_ms','frac_fwd','frac_bwd','frac_fwd_bwd')})
L=d.get('layer'); print('layer', {k: L.get(k) for k in ('ms_per_step','error')} , {k: L[k]['ms_per_step'] for k in ('two_node_layers','dropout_off','no_recompute') if k in L}, L.get('projections'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
echo "== two-rank rehearsal of the whole bench flow (gloo, both ranks on this GPU; a stall dumps the stacks)"
WATCHDOG=150 LIMIT=400 bash tools/rehearse_two_ranks.sh 2>&1 | tail -3 | cut -c1-300
echo "== rocprofv3: attention bench"
bash tools/prof_pmc.sh ${TAG} > /dev/null 2>&1; head -24 gpurun_out/prof_${TAG}/summary.md
echo "== rocprofv3: layer section"
bash tools/prof_layer_pmc.sh ${TAG} > /dev/null 2>&1; head -30 gpurun_out/prof_layer_${TAG}/summary.md
echo "== rocprofv3: projection micro-bench"
bash tools/prof_cmd.sh ${TAG}_lnl python tools/bench_ln_linear.py --iters 10 > /dev/null 2>&1; head -22 gpurun_out/prof_${TAG}_lnl/summary.md
} > gpurun_out/${TAG}/validation.txt 2>&1
tail -120 gpurun_out/${TAG}/validation.txt | cut -c1-400
