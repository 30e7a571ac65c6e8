#!/bin/bash
# final round-3 numbers (code of the last commit): default bench, per-workload benches, ops bench, rocprofv3 stats + PMC of the bench command
OUT=gpurun_out/r3; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err
for wl in M-jag M-targets C2 C3 C4 C5; do timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 > $OUT/bench_final_$wl.json 2>> $OUT/bench_final.err; done
timeout 300 python bench.py --head-dim 64 --steps 20 --warmup 5 --no-layer --no-cpu --no-extra > $OUT/bench_final_d64.json 2>> $OUT/bench_final.err
timeout 600 python tools/bench_ops.py 8192 > $OUT/bench_ops_8192.json 2>> $OUT/bench_final.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3/bench_final*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['roofline_fwd']['avg_launch_ms'],3), round(d['roofline']['avg_launch_ms'],3), round(d['roofline_fwd_bwd']['frac'],3), d['roofline']['bound'], round(d['roofline']['frac'],3))
        if 'layer' in d and d['layer']: l=d['layer']; print('  layer', round(l['ms_per_step'],2), 'nodrop', round(l['dropout_off']['ms_per_step'],2), 'norecompute', round(l['no_recompute']['ms_per_step'],2), 'two-node', round(l['two_node_layers']['ms_per_step'],2))
    except Exception as e: print(f, 'ERR', e)
PY
tail -5 $OUT/bench_final.err
tools/prof_pmc.sh r3c > $OUT/prof_pmc_c.log 2>&1; tail -40 $OUT/prof_pmc_c.log
