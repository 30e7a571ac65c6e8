#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py workload (attention section only): tools/prof_stats.sh <tag> [bench args...]
# -> gpurun_out/prof_<tag>/summary.md
set -u
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export PROF_WARMUP=10
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o r -- python $ROOT/bench.py --no-layer --no-cpu --no-extra $* > $OUT/stats.log 2>&1
python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete; head -30 $OUT/summary.md
