#!/bin/bash
OUT=gpurun_out/r05_call10
mkdir -p $OUT
{
echo "== headline stability (telemetry outside the timed loop)"
for i in 1 2 3 4 5 6; do timeout 300 python bench.py --no-layer --no-cpu --no-extra 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline_fwd_bwd']['frac'],4), d['step_spread'], d['telemetry']['headline'].get('sclk_mhz'), d['telemetry']['headline'].get('samples'))"; done
} > $OUT/log.txt 2>&1
cat $OUT/log.txt | cut -c1-300
for v in v1 v2; do
  export HSTU_BWD_W16=1 HSTU_HIP_LIBRARY=$PWD/tests/probe/libhstu_w16$v.so
  bash tools/prof_pmc.sh r05_w16$v --parity-users 0 > /dev/null 2>&1
  unset HSTU_BWD_W16 HSTU_HIP_LIBRARY
  grep -A40 "hstu_attn_bwd_w16" gpurun_out/prof_r05_w16$v/summary.md | head -8 | cut -c1-200
done
