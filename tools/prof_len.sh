cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p256 -o r -- python $GRAFT_REPO_ROOT/tools/scan_len.py 256 > /tmp/p256.log 2>&1
python - <<'PY'
import glob, sqlite3
for db in glob.glob('/tmp/p256/**/*.db', recursive=True):
    cur = sqlite3.connect(db).cursor()
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 8"):
        print(f"{name[:80]:80s} {calls:5d} {avg/1e3:10.1f} us {pct:5.1f}%")
PY
