#!/usr/bin/env python3
"""Summarise rocprofv3 output dirs written by tools/prof_pmc.sh into a markdown table:
per kernel average duration (kernel trace) and per-dispatch average of every PMC counter."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def short(name):
    for key in ("hstu_ln_linear_fwd_kernel", "Cijk_", "hstu_attn_bwd_dkv_kernel", "hstu_attn_bwd_dq_kernel", "hstu_attn_bwd_fold_bias_kernel", "hstu_attn_bwd_fold_kernel", "hstu_attn_bwd_quad_kernel", "hstu_attn_bwd_solo_bias_kernel", "hstu_attn_fwd_solo_bias_kernel", "hstu_attn_bwd_solo_kernel",
                "hstu_attn_fwd_solo_kernel", "hstu_attn_bwd_kernel", "hstu_attn_fwd_kernel", "hstu_dq_convert", "layer_norm", "norm_mul", "silu"):
        if key in name:
            return key
    return None


def main(root):
    print(f"# rocprofv3 summary of {os.path.basename(root)}\n")
    for db in sorted(glob.glob(os.path.join(root, "stats", "**", "*.db"), recursive=True)):
        cur = sqlite3.connect(db).cursor()
        print("## kernel trace (--kernel-trace --stats)\n\n| kernel | calls | avg us | total us | % |\n|---|---|---|---|---|")
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 8"):
            print(f"| {name[:70]} | {calls} | {avg:.1f} | {total:.0f} | {pct:.1f} |")
        # steady state: the timed launches of bench.py are the LAST `steps` dispatches of each attention kernel (the
        # first ones are its warm-up, at ramping clocks); bench.py's HIP-event figure is over exactly those
        print("\n| kernel | dispatches | avg us, all | avg us, warm-up excluded | median us |\n|---|---|---|---|---|")
        for (name,) in list(cur.execute("select distinct name from kernels where name like '%hstu_attn%' or name like '%hstu_ln_linear%' or name like 'Cijk_%'")):
            d = [r[0] / 1e3 for r in cur.execute("select duration from kernels where name = ? order by start", (name,))]
            warm = int(os.environ.get("PROF_WARMUP", "2"))
            tail = d[warm:] if len(d) > warm else d
            sd = sorted(tail)
            print(f"| {name[:70]} | {len(d)} | {sum(d) / len(d):.1f} | {sum(tail) / len(tail):.1f} | {sd[len(sd) // 2]:.1f} |")
        rows = list(cur.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x from kernels where name like '%hstu_attn%' or name like '%hstu_ln_linear%' or name like 'Cijk_%' group by name"))
        print("\n| kernel | arch VGPR | accum VGPR | SGPR | LDS bytes | grid | block |\n|---|---|---|---|---|---|---|")
        for r in rows:
            print("| " + " | ".join(str(x)[:60] for x in r) + " |")
    agg = defaultdict(lambda: defaultdict(list))
    for db in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*.db"), recursive=True)):
        cur = sqlite3.connect(db).cursor()
        try:
            cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        except Exception as e:
            print("no counters in", db, e)
            continue
        ni, ci, vi = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"), cols.index("counter_name"), cols.index("value")
        di = cols.index("dispatch_id")
        per = defaultdict(float)
        for row in cur.execute("select * from counters_collection"):
            k = short(row[ni])
            if k:
                per[(k, row[di], row[ci])] += row[vi]
        for (k, d, c), v in per.items():
            agg[k][c].append(v)
    print("\n## PMC counters (average per dispatch, summed over XCDs/SEs)\n")
    for k, ctrs in agg.items():
        print(f"### {k}\n\n| counter | avg per dispatch |\n|---|---|")
        for c, vals in sorted(ctrs.items()):
            print(f"| {c} | {sum(vals) / len(vals):.4g} |")
        print()


if __name__ == "__main__":
    main(sys.argv[1])
