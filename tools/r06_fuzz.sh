#!/bin/bash
# full randomised sweeps against the oracle (round-6 final library: masks by tile class, length-class launches, counter hand-out, heavy-first order)
OUT=gpurun_out/r06_fuzz
mkdir -p $OUT
timeout 900 python tools/fuzz_attention.py --cases 500 --seed 70 > $OUT/mha.txt 2>&1
timeout 600 python tools/fuzz_attention.py --cases 40 --seed 71 --big > $OUT/mha_big.txt 2>&1
timeout 900 python tools/fuzz_attention.py --cases 300 --seed 72 --bias > $OUT/bias.txt 2>&1
timeout 900 python tools/fuzz_ops.py --cases 200 --seed 73 > $OUT/ops.txt 2>&1
for f in mha mha_big bias ops; do echo "== $f"; tail -4 $OUT/$f.txt | cut -c1-300; done
