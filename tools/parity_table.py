#!/usr/bin/env python3
"""gpurun_out/parity_errors.json (written by the GPU suite, tests/conftest.py::record_parity) -> the markdown table under
profiles/: per test function and output dtype, the LARGEST measured value over all its checks.
python tools/parity_table.py [in.json] > profiles/rNN_parity_errors.md"""
import json, re, sys
from collections import defaultdict

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_errors.json"
recs = json.load(open(path))
groups = defaultdict(list)
for r in recs:
    name = re.sub(r"\[.*\]$", "", r["test"].replace("tests/", ""))
    groups[(name, r.get("dtype") or "-")].append(r)
ntests = len({r["test"] for r in recs})
print(f"| test | output dtype | checks | max rel. Frobenius | max err / max ref | vs reference rounded to the dtype | fp16: after one subnormal quantum of slack |")
print("|---|---|---|---|---|---|---|")
mx = lambda rs, k: max((r[k] for r in rs if k in r), default=None)
fmt = lambda v: "" if v is None else f"{v:.2e}"
for (name, dt), rs in sorted(groups.items()):
    print(f"| `{name}`  | {dt} | {len(rs)} | {fmt(mx(rs, 'rel_fro'))} | {fmt(mx(rs, 'max_err_over_max_ref'))} | {fmt(mx(rs, 'rel_fro_vs_rounded_ref'))} | {fmt(mx(rs, 'rel_fro_after_fp16_quantum'))} |")
by = defaultdict(list)
for r in recs:
    by[r.get("dtype")].append(r["rel_fro"])
print()
for dt, v in sorted(by.items(), key=lambda kv: str(kv[0])):
    print(f"* {dt}: {len(v)} checks, largest relative Frobenius error {max(v):.3e}")
print(f"\n{len(recs)} checks of {ntests} parametrised tests.")
