#!/bin/bash
OUT=gpurun_out/r3; mkdir -p $OUT
timeout 300 python tools/ab_bwd.py --fwd tests/probe/libhstu_base.so generative_recommenders_amd/libhstu_hip.so > $OUT/ab17.txt 2>&1; cat $OUT/ab17.txt
HSTU_FWD16=0 timeout 300 python tools/ab_bwd.py --fwd generative_recommenders_amd/libhstu_hip.so > $OUT/ab17b.txt 2>&1; cat $OUT/ab17b.txt
timeout 300 python tools/ab_bwd.py --fwd --workload M-jag tests/probe/libhstu_base.so generative_recommenders_amd/libhstu_hip.so 2>&1 | tail -2
timeout 300 python tools/ab_bwd.py --fwd --head-dim 64 tests/probe/libhstu_base.so generative_recommenders_amd/libhstu_hip.so 2>&1 | tail -2
timeout 2000 python -m pytest tests -m gpu -q -x > $OUT/gputest4.txt 2>&1; tail -8 $OUT/gputest4.txt
