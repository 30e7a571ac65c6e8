#!/usr/bin/env python3
"""Assemble profiles/rNN_pmc_traffic.json -- the HBM-traffic figures bench.py attaches to its roofline objects -- from the
summaries of tools/prof_traffic.sh passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes, per dispatch).

    tools/make_pmc_traffic.py --out gpurun_out/r06_pmc_traffic.json \
        --pass M-full:8192:128:4:gpurun_out/prof_traffic_M-full [--pass ...]

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: rocprofiler reports both in units of 1024 bytes, and on gfx950
FETCH_SIZE tallies the 128-byte requests of a wide coalesced stream as 64 bytes (MI355X_MICROARCH.md, section HBM).  Every entry is
stamped with the SHA-256 of the attention kernels' sources (bench.kernel_sources_sha256): bench.py refuses an entry whose stamp is
not that of the sources it runs on."""
import argparse
import importlib.util
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_traffic", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_for_traffic"] = mod
    spec.loader.exec_module(mod)
    return mod


def parse_summary(path):
    """{short kernel name: {counter: average per dispatch}} of a tools/prof_summary.py markdown file"""
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^### (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.match(r"^\| (\w+) \| ([-+0-9.eE]+) \|", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--pass", dest="passes", action="append", default=[], help="workload:users:head_dim:heads:dir-with-summary.md")
    ap.add_argument("--note", default="")
    args = ap.parse_args()
    bench = _bench()
    import torch
    from generative_recommenders_amd.ops import _launch

    sha = bench.kernel_sources_sha256()
    entries = []
    for spec in args.passes:
        wl, users, d, h, pdir = spec.split(":", 4)
        users, d, h = int(users), int(d), int(h)
        summ = parse_summary(os.path.join(pdir, "summary.md"))
        n = bench.WORKLOADS[wl][0]
        bias = wl in ("C2", "C3-bias")
        names = {"fwd": _launch.attn_fwd_kernel_name(torch.bfloat16, d, d, n, heads=h, alpha=d ** -0.5, with_bias=bias)}
        if wl != "C5":
            names["bwd"] = _launch.attn_bwd_kernel_name(torch.bfloat16, d, d, n, heads=h, alpha=d ** -0.5, with_bias=bias)
        ent = {"source": {"workload": wl, "users_per_gpu": users, "head_dim": d, "heads": h, "dtype": "bf16"}, "sources_sha256": sha}
        for side, full in names.items():
            # ("a<..>+b<..>": a path of two kernels per call -- the long-sequence backward -- counts the bytes of both)
            c, missing = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}, False
            for part in full.split("+"):
                short = part.split("<")[0]
                cand = [k for k in summ if k == short] or [k for k in summ if short.startswith(k) or k.startswith(short)]
                if not cand or "FETCH_SIZE" not in summ[cand[0]] or "WRITE_SIZE" not in summ[cand[0]]:
                    print(f"{spec}: no FETCH_SIZE / WRITE_SIZE for {part} (sections: {sorted(summ)})", file=sys.stderr)
                    missing = True
                    break
                c["FETCH_SIZE"] += summ[cand[0]]["FETCH_SIZE"]
                c["WRITE_SIZE"] += summ[cand[0]]["WRITE_SIZE"]
            if missing:
                continue
            # the short-sequence backward kernels run as TWO dispatches per step (length classes, round 6): the summary's average per
            # dispatch is half a step's bytes
            if side == "bwd" and "solo" in full and os.environ.get("HSTU_SOLO_SPLIT", "1") != "0" and n > 32:
                c = {k: 2.0 * v for k, v in c.items()}
            ent[side] = {"kernel": full, "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"],
                         "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
                         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, average per dispatch; bytes = (2 x FETCH_SIZE + "
                                 "WRITE_SIZE) x 1024 (gfx950 correction of MI355X_MICROARCH.md); summary: " + os.path.join(pdir, "summary.md")}
        if "fwd" in ent:
            entries.append(ent)
    json.dump({"sources_sha256": sha, "globs": list(bench.KERNEL_SOURCE_GLOBS), "note": args.note, "entries": entries}, open(args.out, "w"), indent=1)
    print(f"{args.out}: {len(entries)} entries, sources {sha[:12]}")


if __name__ == "__main__":
    main()
