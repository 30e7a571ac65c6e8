#!/bin/bash
# HBM-traffic passes of every bench workload on the current kernel sources (-> tools/make_pmc_traffic.py -> profiles/r06_pmc_traffic.json)
bash tools/prof_traffic.sh traffic_M-full > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_M-jag --workload M-jag > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_M-targets --workload M-targets --sort-by-length 1 > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_M-full-1024 --users-per-gpu 1024 > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_M-full-d64 --head-dim 64 > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_C2 --workload C2 > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_C3 --workload C3 > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_C3-bias --workload C3-bias > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_C4 --workload C4 > /dev/null 2>&1
for d in gpurun_out/prof_traffic_*; do echo "$d $(grep -c SIZE $d/summary.md)"; done
