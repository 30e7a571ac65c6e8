#!/bin/bash
# (record of the round's first GPU call; the sixteen-wave kernel it exercises has since left the library: docs/experiments/r05_*, HSTU_BWD_W16 is a no-op now)
# Round-5 first GPU call: (1) the new glue ops + the layer tests, (2) the sixteen-wave backward against the tests of the folded
# one, (3) A/B timing fold vs w16 on the metric shape (+ ablations), (4) the default bench line with the new sections.
OUT=gpurun_out/r05_call1
mkdir -p $OUT
{
echo "== glue + compute + ln_linear + abi tests"
timeout 900 python -m pytest tests/test_glue_gpu.py tests/test_compute_gpu.py tests/test_ln_linear_gpu.py tests/test_abi.py -q -m gpu -x 2>&1 | tail -15
echo "== w16: backward tests with HSTU_BWD_W16=1"
HSTU_BWD_W16=1 timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_metric_shapes_gpu.py -q -m gpu -x -k "not test_the_headline_backward" 2>&1 | tail -15
echo "== w16: kernel name"
HSTU_BWD_W16=1 python -c "
import torch
from generative_recommenders_amd.ops import _launch
print(_launch.attn_bwd_kernel_name(torch.bfloat16, 128, 128, 200))"
echo "== A/B: fold"
HSTU_BWD_W16=0 timeout 300 python bench.py --no-layer --no-cpu --no-extra --steps 30 --warmup 10 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fold', d['roofline']['kernel'], 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'parity', d['parity_at_this_size'])"
echo "== A/B: w16"
HSTU_BWD_W16=1 timeout 300 python bench.py --no-layer --no-cpu --no-extra --steps 30 --warmup 10 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('w16', d['roofline']['kernel'], 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'parity', d['parity_at_this_size'])"
for v in 96 64 32; do
  if [ -f tests/probe/libhstu_w16a${v}.so ]; then
    echo "== w16 ablation ${v}"
    HSTU_BWD_W16=1 HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_w16a${v}.so timeout 300 python bench.py --no-layer --no-cpu --no-extra --steps 20 --warmup 5 --parity-users 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('w16 ablate', d['roofline']['kernel'], 'bwd', d['roofline']['avg_launch_ms'])"
  fi
done
echo "== bench default"
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_call1/bench_default.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'fwd', d['roofline_fwd']['avg_launch_ms'], d['roofline_fwd']['frac'], 'bwd', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'both', d['roofline_fwd_bwd']['frac'])
print('parity', d['parity_at_this_size'])
print('telemetry', json.dumps(d.get('telemetry')))
print('calibration', d.get('calibration'))
L=d.get('layer'); print('layer', {k: L.get(k) for k in ('ms_per_step','error')}, {k: L[k]['ms_per_step'] for k in ('two_node_layers','dropout_off','no_recompute') if k in L})
for k,v in (L.get('projections') or {}).items(): print('   ', k, v)
for k,v in d['extra_workloads'].items(): print(' ', k, {x: v.get(x) for x in ('fwd_ms','bwd_ms','frac_fwd','frac_bwd','frac_fwd_bwd','error')})
PY
echo "== layer: dbeta stream A/B"
for m in 1 0; do HSTU_DBETA_STREAM=$m timeout 300 python bench.py --no-cpu --no-extra --steps 5 --warmup 2 --parity-users 0 2>&1 | tail -1 | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); L=d['layer']; print('dbeta_stream', os.environ.get('HSTU_DBETA_STREAM'), L.get('ms_per_step'), L.get('error'))"; done
} > $OUT/log.txt 2>&1
tail -150 $OUT/log.txt | cut -c1-600
