#!/bin/bash
# HBM traffic of one bench.py workload: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (+ kernel trace),
# summarised by tools/prof_summary.py.  tools/prof_traffic.sh <tag> [bench args...] -> gpurun_out/prof_<tag>/summary.md
set -u
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 6 --warmup 2 --no-layer --no-cpu --no-extra $*"
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/pmc_$i -o r -- $BENCH > $OUT/pmc_$i.log 2>&1
done
python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.md 2>&1
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete; cat $OUT/summary.md
