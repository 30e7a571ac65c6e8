#!/bin/bash
OUT=gpurun_out/r05_call7
mkdir -p $OUT
{
for m in 0 1; do
  if [ $m = 1 ]; then F="--layer-tunableop"; else F=""; fi
  timeout 600 python bench.py --no-cpu --no-extra --steps 10 --warmup 5 --parity-users 0 $F 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); L=d['layer']; print('tunableop=$m layer', L.get('ms_per_step'), L.get('error'), L.get('gemm_selection'), {k: L[k]['ms_per_step'] for k in ('two_node_layers','dropout_off','no_recompute') if k in L}); print({k:(v['us'],v['mfma_frac']) for k,v in L['projections'].items()})"
done
ls -la tunableop_results*.csv 2>/dev/null; cp tunableop_results*.csv $OUT/ 2>/dev/null
} > $OUT/log.txt 2>&1
cat $OUT/log.txt | cut -c1-900
