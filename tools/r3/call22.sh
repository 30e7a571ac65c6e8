#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_r03b_swaps; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
HSTU_HIP_LIBRARY=$ROOT/tests/probe/libhstu_swaps.so timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE -d $OUT/pmc_3 -o r -- python $ROOT/bench.py --steps 6 --warmup 2 --no-layer --no-cpu --no-extra > $OUT/pmc_3.log 2>&1
cd $ROOT
python tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
tail -30 $OUT/summary.md
