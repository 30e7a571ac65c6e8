#!/bin/bash
# (record; needs the experiment's library builds -- see docs/EXPERIMENTS.md R5.1 for how they were made from commit 17e3774)
# A/B of the sixteen-wave backward against the folded one on the metric shape (+ its ablation builds), correctness first
OUT=gpurun_out/r05_w16_ab
mkdir -p $OUT
{
echo "== w16: backward tests with HSTU_BWD_W16=1"
HSTU_BWD_W16=1 timeout 900 python -m pytest tests/test_attention_gpu.py -q -m gpu -x -k "fold or batch_composition or strided or golden" 2>&1 | tail -6
one() { # name env...
  env "$@" timeout 300 python bench.py --no-layer --no-cpu --no-extra --steps 30 --warmup 10 --parity-users ${PU:-32} 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['parity_at_this_size']; print('$1', d['roofline']['kernel'], 'fwd', round(d['roofline_fwd']['avg_launch_ms'],4), 'bwd', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), 'parity', p.get('max_rel_fro') if isinstance(p, dict) else None, p.get('ok') if isinstance(p, dict) else None)"
}
one HSTU_BWD_W16=0
one HSTU_BWD_W16=1
for v in 96 64 32; do
  [ -f tests/probe/libhstu_w16a${v}.so ] && PU=0 one HSTU_BWD_W16=1 HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_w16a${v}.so
done
echo "== M-jag / 1024 users"
for w in 0 1; do
HSTU_BWD_W16=$w timeout 300 python bench.py --no-layer --no-cpu --no-extra --workload M-jag --steps 30 --warmup 10 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('M-jag w16=$w', d['roofline']['kernel'], 'bwd', round(d['roofline']['avg_launch_ms'],4), d['parity_at_this_size'].get('max_rel_fro'))"
HSTU_BWD_W16=$w timeout 300 python bench.py --no-layer --no-cpu --no-extra --users-per-gpu 1024 --steps 96 --warmup 20 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('M-full-1024 w16=$w', 'bwd', round(d['roofline']['avg_launch_ms'],4), d['parity_at_this_size'].get('max_rel_fro'))"
done
} > $OUT/log.txt 2>&1
tail -60 $OUT/log.txt | cut -c1-400
