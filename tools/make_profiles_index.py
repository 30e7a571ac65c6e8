#!/usr/bin/env python3
"""profiles/INDEX.md: one line per evidence file -- what it holds (its own first line / JSON keys), the commit that added it.
Every file is the output of ONE gpurun call, i.e. one fresh single-GPU MI355X box, unless the file says otherwise."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def first_commit(path):
    out = subprocess.run(["git", "log", "--diff-filter=A", "--format=%h %ad", "--date=short", "--", path], cwd=ROOT,
                         capture_output=True, text=True).stdout.strip().splitlines()
    return out[-1] if out else "(uncommitted)"


def describe(path):
    try:
        if path.endswith(".json"):
            txt = open(path).read().strip()
            try:
                d = json.loads(txt) if txt else {}
            except ValueError:
                d = json.loads(txt.splitlines()[-1])
            if isinstance(d, dict):
                if "metric" in d and "value" in d:
                    r = d.get("roofline", {}) or {}
                    wl = str((d.get("config") or {}).get("workload", ""))[:70]
                    return f"bench.py line: {wl}; value {d['value']:.4g} {d.get('unit', '')}, dominant kernel frac {r.get('frac', 0):.3f}"
                return "JSON: " + ", ".join(list(d.keys())[:8])
            return "JSON list"
        for line in open(path, errors="replace"):
            t = line.strip().lstrip("#").strip()
            if t and not t.startswith("|--") and "Warning" not in t:
                return t[:200]
    except Exception as e:  # noqa: BLE001
        return f"(unreadable: {e})"
    return ""


def main():
    rows = []
    for name in sorted(os.listdir(P)):
        if name == "INDEX.md":
            continue
        path = os.path.join(P, name)
        if os.path.isdir(path):
            continue
        rows.append((name, describe(path), first_commit(os.path.join("profiles", name))))
    with open(os.path.join(P, "INDEX.md"), "w") as f:
        f.write("# profiles/ -- index of the evidence files\n\n"
                "One line per file: what it holds (the file's own first line, or the headline of a bench line), and the commit that added it.\n"
                "Files are named `rNN_*` by round.  Each is the output of one `gpurun` call = one fresh single-GPU MI355X box (numbers of\n"
                "different files are numbers of different boxes unless a file says that it is a same-process A/B).  Regenerate with\n"
                "`python tools/make_profiles_index.py`.\n\n| file | what | added by |\n|---|---|---|\n")
        for n, d, c in rows:
            f.write(f"| `{n}` | {d.replace('|', '/')} | {c} |\n")
    print(len(rows), "files indexed")


if __name__ == "__main__":
    main()
