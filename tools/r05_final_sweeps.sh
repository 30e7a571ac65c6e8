#!/bin/bash
# final library: PMC passes of the research-path (C2) kernels + the randomised sweeps that touch code changed late in the round
OUT=gpurun_out/r05_final_sweeps
mkdir -p $OUT
bash tools/prof_pmc.sh r05_c2 --workload C2 --parity-users 0 > /dev/null 2>&1
timeout 700 python tools/fuzz_attention.py --cases 250 --seed 61 --bias > $OUT/bias.txt 2>&1
timeout 700 python tools/fuzz_ops.py --cases 200 --seed 62 > $OUT/ops.txt 2>&1
timeout 500 python tools/fuzz_attention.py --cases 200 --seed 63 > $OUT/mha.txt 2>&1
for f in bias ops mha; do echo "== $f"; grep -i -E "cases|failures" $OUT/$f.txt | head -5; done
grep -n "^###" gpurun_out/prof_r05_c2/summary.md
