#!/usr/bin/env python3
"""Static instruction census of the kernels in a gfx950 assembly file (hipcc -save-temps: *-gfx950.s).

Per kernel: SGPR / VGPR counts from the metadata, and the number of VALU, SALU, MFMA, LDS, VMEM, s_waitcnt instructions,
`v_readlane_b32` / `v_writelane_b32` (SGPR values parked in VGPR lanes: the register allocator ran out of scalar registers)
and `s_load` (kernel-argument re-reads).  CPU only.  Usage: tools/isa_census.py file.s [name-filter]"""
import re
import sys
from collections import Counter


def census(path, filt=None):
    out = {}
    cur = None
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur = m.group(1)
            out[cur] = Counter()
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            m2 = re.match(r";\s*(SGPRBlocks|NumSgprs|NumVgprs|NumAgprs|ScratchSize|Occupancy|TotalNumSgprs):\s*(\d+)", s)
            continue
        op = s.split()[0]
        c = out[cur]
        c["total"] += 1
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            c["readlane" if op.startswith("v_readlane") else "readfirstlane"] += 1
            c["valu"] += 1
        elif op.startswith("v_writelane"):
            c["writelane"] += 1
            c["valu"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            c["s_load"] += 1
        elif op.startswith("s_barrier"):
            c["barrier"] += 1
        elif op.startswith("s_nop"):
            c["s_nop"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
            if op.startswith("scratch_"):
                c["scratch"] += 1
    # metadata
    meta = {}
    txt = open(path).read()
    for blk in txt.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
        meta[name.group(1)] = dict(sgpr=g("sgpr_count"), vgpr=g("vgpr_count"), spill=g("vgpr_spill_count"),
                                   sspill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"))
    rows = []
    for k, c in out.items():
        if k not in meta or (filt and filt not in k):
            continue
        rows.append((k, meta[k], c))
    return rows


if __name__ == "__main__":
    rows = census(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    for k, m, c in rows:
        print(k)
        print("   sgpr %d vgpr %d vgpr_spill %d sgpr_spill %d scratch %d" % (m["sgpr"], m["vgpr"], m["spill"], m["sspill"], m["scratch"]))
        print("   " + "  ".join(f"{n} {c[n]}" for n in ("total", "valu", "salu", "mfma", "lds", "vmem", "waitcnt", "s_load", "readlane", "writelane", "readfirstlane", "s_nop", "barrier", "scratch")))
