#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$OUT/tunableop.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=200 timeout 900 python bench.py --no-cpu --no-extra --steps 10 --warmup 3 > $OUT/bench_tunable.json 2> $OUT/bench_tunable.err ) 2>&1 | tail -3
( time timeout 900 python bench.py --no-cpu --no-extra --steps 10 --warmup 3 > $OUT/bench_notune.json 2> $OUT/bench_notune.err ) 2>&1 | tail -3
python - <<'PY'
import json
for n in ('tunable','notune'):
    d=json.loads(open(f'gpurun_out/r3/bench_{n}.json').read().strip().splitlines()[-1])
    l=d['layer']; print(n, l.get('ms_per_step'), l.get('dropout_off'), l.get('no_recompute'), {k:v['mfma_frac'] for k,v in l.get('projections',{}).items()})
PY
ls -la $OUT/tunableop*.csv 2>/dev/null; head -20 $OUT/tunableop0.csv 2>/dev/null; tail -3 $OUT/bench_tunable.err
