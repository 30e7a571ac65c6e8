#!/usr/bin/env python3
"""Per-kernel table of the layer profile (tools/prof_layer_pmc.sh): average duration (kernel trace) and, from the PMC passes,
the MFMA pipe's busy fraction (projections) and the achieved HBM rate (row kernels)."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def key(name):
    if name.startswith("Cijk") or name.startswith("Custom_Cijk"):
        return "GEMM " + name.split("_MT")[0][:40] + " MT" + name.split("_MT")[1].split("_")[0] if "_MT" in name else "GEMM " + name[:60]
    for k in ("hstu_ln_linear_fwd_kernel", "hstu_attn_bwd_fold_kernel", "hstu_attn_fwd_kernel", "norm_mul_fwd_gn_kernel", "norm_mul_bwd_gn_kernel", "layer_norm_fwd_kernel",
              "layer_norm_bwd_kernel", "reduce_partials_kernel", "reduce_kernel", "multi_tensor_apply", "copyBuffer", "elementwise"):
        if k in name:
            return k
    return None


def main(root):
    dur = defaultdict(list)
    for db in glob.glob(os.path.join(root, "stats", "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db).cursor()
        for name, d in cur.execute("select name, duration from kernels"):
            k = key(name)
            if k:
                dur[k].append(d / 1e3)
    ctr = defaultdict(lambda: defaultdict(list))
    for db in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*.db"), recursive=True)):
        cur = sqlite3.connect(db).cursor()
        try:
            cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        except Exception:
            continue
        ni = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name")
        ci, vi, di = cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
        per = defaultdict(float)
        for row in cur.execute("select * from counters_collection"):
            k = key(row[ni])
            if k:
                per[(k, row[di], row[ci])] += row[vi]
        for (k, d, c), v in per.items():
            ctr[k][c].append(v)
    tot = sum(sum(v) for v in dur.values())
    print(f"# rocprofv3 of the layer section ({os.path.basename(root)}): 3 STU layers fwd + bwd, 1024 users, dropout 0.1, recompute on\n")
    print("MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x cycles of the dispatch), cycles = GRBM_GUI_ACTIVE / 8 "
          "(it is summed over the 8 XCDs).  HBM GB/s = (2 x FETCH_SIZE + WRITE_SIZE) KiB / duration: FETCH_SIZE doubled as "
          "MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads on gfx950 (calibrated on the attention kernels, whose "
          "traffic then equals their algorithmic bytes; uncalibrated for the GEMMs' access pattern: read their HBM column as an "
          "upper bound).  Averages per dispatch over all the section's launches (warm-up included).\n")
    print("| kernel | calls | avg us | % of kernel time | MFMA pipe busy | FETCH_SIZE avg | WRITE_SIZE avg | HBM GB/s (2 x FETCH + WRITE, KiB) |\n|---|---|---|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        avg = sum(v) / len(v)
        c = ctr.get(k, {})
        mf = ""
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
            act = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
            mf = f"{busy / (4 * 256 * act / 8):.3f}" if act else ""       # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) if "FETCH_SIZE" in c else None
        w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) if "WRITE_SIZE" in c else None
        hbm = f"{(2 * f + w) * 1024 / (avg * 1e-6) / 1e9:.0f}" if f is not None and w is not None else ""
        print(f"| {k} | {len(v)} | {avg:.1f} | {100 * sum(v) / tot:.1f} | {mf} | {'' if f is None else f'{f:.4g}'} | {'' if w is None else f'{w:.4g}'} | {hbm} |")


if __name__ == "__main__":
    main(sys.argv[1])
