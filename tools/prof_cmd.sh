#!/bin/bash
# rocprofv3 kernel-trace stats + PMC counters (separate passes, as gpurun requires) of an arbitrary command.
# Usage (on the GPU box): tools/prof_cmd.sh <tag> <command...>   -> gpurun_out/prof_<tag>/summary.md (tools/prof_summary.py)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export PROF_WARMUP=${PROF_WARMUP:-5}
( cd $ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o r -- "$@" > $OUT/stats.log 2>&1 )
i=0
for CTRS in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
  "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  ( cd $ROOT && timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/pmc_$i -o r -- "$@" > $OUT/pmc_$i.log 2>&1 )
done
python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete; cat $OUT/summary.md
