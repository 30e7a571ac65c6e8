#!/bin/bash
# kernel-level trace of tools/bench_research.py (research-path attention with the fused relative bias)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_research
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_research -o r -- python $ROOT/tools/bench_research.py > /dev/null 2>&1
cd $ROOT
python - <<PY
import sqlite3, glob
db=glob.glob("gpurun_out/prof_research/**/*.db", recursive=True)[0]
cur=sqlite3.connect(db).cursor()
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 8"):
    print(f"{name[:84]:84s} {calls:5d} {avg/1e6:9.3f} ms {pct:5.1f}%")
PY
