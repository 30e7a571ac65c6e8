#!/bin/bash
# (record) A/B of s_setprio by phase (FOLD_SETPRIO / FWD_SETPRIO builds: tools/build_variant.sh) against the default build, one process per
# kernel, variants interleaved round-robin (tools/ab_bwd.py)
OUT=gpurun_out/r05_setprio
mkdir -p $OUT
P=tests/probe
{
for wl in M-full M-jag; do
echo "== backward $wl"
timeout 300 python tools/ab_bwd.py --workload $wl --reps 7 $P/libhstu_base0.so $P/libhstu_sp1.so $P/libhstu_sp2.so $P/libhstu_sp4.so
echo "== forward $wl"
timeout 300 python tools/ab_bwd.py --workload $wl --reps 7 --fwd $P/libhstu_base0.so $P/libhstu_fp1.so $P/libhstu_fp2.so $P/libhstu_fp4.so $P/libhstu_fp9.so $P/libhstu_fp10.so $P/libhstu_fp12.so
done
} > $OUT/log.txt 2>&1
tail -50 $OUT/log.txt | cut -c1-300
