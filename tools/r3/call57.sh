#!/bin/bash
mkdir -p gpurun_out/r3
T0=$(date +%s); python bench.py > gpurun_out/r3/bench_g_default.json 2> gpurun_out/r3/bench_g_default.err; echo "wall $(( $(date +%s) - T0 )) s"

python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_g_default.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'], d['roofline_fwd_bwd']['frac'])
for k,v in d['extra_workloads'].items(): print(' ', k, round(v['user_seqs_per_s']), v['fwd_ms'], v['bwd_ms'], v['frac_fwd_bwd'])
l=d['layer']; print(' layer', round(l['ms_per_step'],2), round(l['dropout_off']['ms_per_step'],2), round(l['no_recompute']['ms_per_step'],2), round(l['two_node_layers']['ms_per_step'],2))
PY
