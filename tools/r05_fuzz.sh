#!/bin/bash
# full randomised sweeps against the oracle (round-5 library): outputs under gpurun_out/r05_fuzz/
OUT=gpurun_out/r05_fuzz
mkdir -p $OUT
timeout 900 python tools/fuzz_attention.py --cases 400 --seed 50 > $OUT/mha.txt 2>&1
timeout 600 python tools/fuzz_attention.py --cases 30 --seed 51 --big > $OUT/mha_big.txt 2>&1
timeout 900 python tools/fuzz_attention.py --cases 250 --seed 52 --bias > $OUT/bias.txt 2>&1
timeout 900 python tools/fuzz_ops.py --cases 200 --seed 53 > $OUT/ops.txt 2>&1
for f in mha mha_big bias ops; do echo "== $f"; tail -4 $OUT/$f.txt | cut -c1-300; done
