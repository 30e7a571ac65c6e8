#!/bin/bash
# Collect rocprofv3 kernel-trace stats + PMC counters (separate passes, as gpurun requires)
# for the attention bench.  Usage (on the GPU box): tools/prof_pmc.sh <tag> [bench args...]
# Output: gpurun_out/prof_<tag>/{stats,pmc_*}/..., summarised by tools/prof_summary.py
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 6 --warmup 2 --no-layer --no-cpu --no-extra $*"
# kernel trace of the SAME command the bench line comes from (default --steps 50 --warmup 10, attention section only)
export PROF_WARMUP=10
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o r -- python $ROOT/bench.py --no-layer --no-cpu --no-extra $* > $OUT/stats.log 2>&1
i=0
for CTRS in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/pmc_$i -o r -- $BENCH > $OUT/pmc_$i.log 2>&1
done
python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete; cat $OUT/summary.md
