#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputest1.txt 2>&1; tail -5 $OUT/gputest1.txt
timeout 600 python bench.py > $OUT/bench1.json 2> $OUT/bench1.err; tail -c 1500 $OUT/bench1.json
