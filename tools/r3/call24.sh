#!/bin/bash
# final round-3 numbers: default bench (with extras + layer + cpu), per-workload benches, ops bench
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
    except Exception as e: print(f, 'ERR', e)
PY
tail -5 $OUT/bench_final.err
