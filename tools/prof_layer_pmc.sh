#!/bin/bash
# rocprofv3 of the LAYER section (3 STU layers fwd + bwd, 1024 users, bench.py's `layer`): kernel trace + PMC passes for
#  - the projections (hipBLASLt GEMMs): MFMA pipe busy cycles vs the kernel's duration  -> MFMA utilisation from counters
#  - the row kernels (norm / SiLU-fused gating / reductions): FETCH_SIZE + WRITE_SIZE     -> achieved HBM GB/s from counters
# Usage (GPU box): tools/prof_layer_pmc.sh <tag>   ->  gpurun_out/prof_layer_<tag>/summary.md
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_layer_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
RUN="python $ROOT/tools/prof_layer_run.py"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o r -- $RUN > $OUT/stats.log 2>&1
i=0
for CTRS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/pmc_$i -o r -- $RUN > $OUT/pmc_$i.log 2>&1
done
python $ROOT/tools/prof_layer_summary.py $OUT > $OUT/summary.md 2>&1
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete; cat $OUT/summary.md
