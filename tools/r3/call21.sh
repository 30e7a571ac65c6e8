#!/bin/bash
# round-3 profiles: rocprofv3 kernel trace + PMC passes of the default bench command (attention section)
mkdir -p gpurun_out/r3
tools/prof_pmc.sh r03b > gpurun_out/r3/prof_r03b.log 2>&1
# the same LDS-conflict pass with the park-swap variants switched on (evidence that the remaining conflicts are the park writes)
OUT=gpurun_out/prof_r03b_swaps; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
HSTU_HIP_LIBRARY=$GRAFT_REPO_ROOT/tests/probe/libhstu_swaps.so timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE -d $OUT/pmc_3 -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-layer --no-cpu --no-extra > $OUT/pmc_3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
tail -70 gpurun_out/prof_r03b/summary.md; tail -30 $OUT/summary.md
