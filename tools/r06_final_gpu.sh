#!/bin/bash
# Round 6, end-of-round GPU pass 1: whole GPU suite, smoke, HBM-traffic passes of every bench workload (for profiles/r06_pmc_traffic.json),
# rocprofv3 kernel trace + PMC of the attention bench, of C2 and of C3-bias, of the layer section.  Everything lands under gpurun_out/.
TAG=r06_final
mkdir -p gpurun_out/${TAG}
{
echo "== pytest -m gpu (all)"
timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -12
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== traffic passes"
bash tools/prof_traffic.sh traffic_M-full > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_M-jag --workload M-jag > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_M-targets --workload M-targets --sort-by-length 1 > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_M-full-1024 --users-per-gpu 1024 > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_M-full-d64 --head-dim 64 > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_C2 --workload C2 > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_C3 --workload C3 > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_C3-bias --workload C3-bias > /dev/null 2>&1
bash tools/prof_traffic.sh traffic_C4 --workload C4 > /dev/null 2>&1
for d in gpurun_out/prof_traffic_*; do echo "-- $d"; grep -A4 "^### " $d/summary.md | head -24; done
echo "== rocprofv3: attention bench (M-full)"
bash tools/prof_pmc.sh ${TAG} > /dev/null 2>&1; head -40 gpurun_out/prof_${TAG}/summary.md
echo "== rocprofv3: C2"
bash tools/prof_pmc.sh ${TAG}_C2 --workload C2 > /dev/null 2>&1; head -40 gpurun_out/prof_${TAG}_C2/summary.md
echo "== rocprofv3: C3-bias"
bash tools/prof_pmc.sh ${TAG}_C3bias --workload C3-bias > /dev/null 2>&1; head -40 gpurun_out/prof_${TAG}_C3bias/summary.md
echo "== rocprofv3: layer section"
bash tools/prof_layer_pmc.sh ${TAG} > /dev/null 2>&1; head -30 gpurun_out/prof_layer_${TAG}/summary.md
} > gpurun_out/${TAG}/validation.txt 2>&1
tail -60 gpurun_out/${TAG}/validation.txt | cut -c1-300
