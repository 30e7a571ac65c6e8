"""Register / scratch budgets of the built kernels, read from the code objects inside libhstu_hip.so (CPU only: the
metadata is in the ELF notes).  The hot kernels sit on occupancy cliffs -- the forward at 3 waves per SIMD (<= 168
VGPRs), the folded backward at 2 (<= 256) -- and a spill or a few extra registers costs tens of percent silently;
this is the regression guard."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "generative_recommenders_amd", "libhstu_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(data):
    """gfx950 ELF images of every clang offload bundle in the library (one bundle per translation unit)."""
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return
        pos = i + len(MAGIC)
        (n,) = struct.unpack_from("<Q", data, pos)
        p = pos + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", data, p)
            ident = data[p + 24: p + 24 + idlen].decode()
            p += 24 + idlen
            if "gfx950" in ident and size:
                yield data[i + off: i + off + size]


def _kernels():
    if not (os.path.exists(LIB) and os.path.exists(READELF)):
        pytest.skip("library or llvm-readelf not available")
    out = {}
    data = open(LIB, "rb").read()
    for co in _code_objects(data):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for block in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            if not name:
                continue
            get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", block).group(1))
            out[name.group(1)] = dict(vgpr=get("vgpr_count"), spill=get("vgpr_spill_count"),
                                      scratch=get("private_segment_fixed_size"), lds=get("group_segment_fixed_size"))
    return out


def test_no_kernel_spills_or_uses_scratch():
    ks = _kernels()
    assert len(ks) > 50, f"only {len(ks)} kernels found in {LIB}"
    # known and accepted: the fp32-I/O general backward with 128-wide values (fragments twice as wide as the 16-bit ones;
    # it is the parity / fp32-user path, not a measured one) spills at the 256-register limit: a few dozen registers at
    # 128 x 128, one at 64 x 128.  The research-path folded backward keeps 4 registers of per-thread offsets in scratch
    # across its user / head loops (written once, reloaded outside the pair loop): recomputing them per problem instead
    # removes the spills and was measured 4 % SLOWER, so the 4 are accepted -- and bounded here.
    accepted = ("hstu_attn_bwd_kernelIfLi128ELi128E", "hstu_attn_bwd_kernelIfLi64ELi128ELb0E")
    # The general research-path backward at 128 x 128 (no research configuration has heads that wide) keeps one.
    # The folded backward at 128 x 128 spills 2 registers since its dQ GEMM got exact slot counts per step (four more copies
    # of that phase): measured 1.4 % FASTER than the spill-free static-slot version on the same box (tools/ab_bwd.py).
    bounded = {"hstu_attn_bwd_fold_bias_kernel": 4, "hstu_attn_bwd_fold_kernelIDF16bLi128ELi128E": 2, "hstu_attn_bwd_fold_kernelIDF16_Li128ELi128E": 2, "hstu_attn_bwd_kernelIDF16bLi128ELi128ELb1E": 1, "hstu_attn_bwd_kernelIDF16_Li128ELi128ELb1E": 1}
    # The WIDE instances of the row kernels (namespace nw4: rows of 1025 .. 4096 elements, csrc/norm_kernels.inc) hold four
    # times the pieces per lane; their backward kernels spill.  They exist so that H x hidden_dim > 1024 works at all (the
    # reference's kernels take any D); every BASELINE configuration (<= 1024) runs the narrow instances, which may not spill.
    accepted = accepted + ("N4hstu3nw4",)
    # u * GroupNorm(attn) backward with SiLU applied on the fly (single-chunk rows): 98 registers, asked to fit the 96 of a
    # fifth wave per SIMD -- 2 spilled registers, measured 4 % faster than 4 waves without (profiles/r03_ab_row_passes_silu.txt);
    # 6 since the sigmoid goes through the hardware exp2 / rcp (round 5): layer step 13.36-13.42 ms against 13.41-13.47 for
    # 4 waves without a spill, three alternating runs (profiles/r05_norm_w5_ab.txt)
    bounded.update({"norm_mul_bwd_gn_kernelIDF16bLi8ELi1ELb1E": 6, "norm_mul_bwd_gn_kernelIDF16_Li8ELi1ELb1E": 6})
    # The short-sequence research backward (one workgroup per CU, one wave per SIMD: 344 registers incl. AGPRs) parks one
    # 8-byte value in scratch at entry (no register spilled in the loops' bodies)
    bounded.update({"hstu_attn_bwd_solo_bias_kernel": 0})
    # The fused LayerNorm + projection kernel's instantiation that also writes the normalised rows (backward's recompute): 8
    # registers of its extra addressing live in scratch across the row prologue; the forward's instantiation may not spill.
    bounded.update({"hstu_ln_linear_fwd_kernelIDF16bLb1E": 8, "hstu_ln_linear_fwd_kernelIDF16_Lb1E": 8})
    # ... and the instantiation without the LayerNorm (hstu_linear_k512) keeps 2 outside the tile loop
    bounded.update({"hstu_ln_linear_fwd_kernelIDF16bLb0ELb0E": 2, "hstu_ln_linear_fwd_kernelIDF16_Lb0ELb0E": 2})
    bad = {k: v for k, v in ks.items() if (v["spill"] or v["scratch"]) and "hstu" in k and not any(a in k for a in accepted)
           and not any(b in k and v["spill"] <= n for b, n in bounded.items())}
    assert not bad, f"kernels with register spills / scratch: {bad}"


def test_hot_kernels_stay_under_their_occupancy_limits():
    ks = _kernels()
    find = lambda frag: [v for k, v in ks.items() if frag in k]
    fwd = find("hstu_attn_fwd_kernelIDF16bLi128ELi128ELb0ELb0ELb0E")   # bf16, 128 x 128, no bias, not the precise variant
    fold = find("hstu_attn_bwd_fold_kernelIDF16bLi128ELi128E")
    fold64 = find("hstu_attn_bwd_fold_kernelIDF16bLi64ELi64E")
    assert len(fwd) == 1 and len(fold) == 1 and len(fold64) == 1
    assert fwd[0]["vgpr"] <= 168, fwd            # 3 waves per SIMD (512 / 3, allocation granule 8)
    assert fold[0]["vgpr"] <= 256, fold          # 2 waves per SIMD
    assert fold64[0]["vgpr"] <= 256, fold64
    lnl = find("hstu_ln_linear_fwd_kernelIDF16bLb0ELb1E")                # two waves per SIMD, the rows of x in 128 of the registers
    assert len(lnl) == 1 and lnl[0]["vgpr"] <= 256 and lnl[0]["spill"] == 0, lnl


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _lane_spill_census(fragments):
    """v_readlane_b32 + v_writelane_b32 per kernel whose mangled name contains one of `fragments` (disassembly of the code objects
    inside the library: scalar values parked in vector lanes show up as exactly these two instructions)."""
    if not (os.path.exists(LIB) and os.path.exists(OBJDUMP)):
        pytest.skip("library or llvm-objdump not available")
    out = {}
    data = open(LIB, "rb").read()
    for co in _code_objects(data):
        if not any(f.encode() in co for f in fragments):
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1) if any(f in m.group(1) for f in fragments) else None
                if cur:
                    out.setdefault(cur, [0, 0])
            elif cur and "v_readlane_b32" in line:
                out[cur][0] += 1
            elif cur and "v_writelane_b32" in line:
                out[cur][1] += 1
    return out


def test_scalar_registers_parked_in_vector_lanes_stay_bounded():
    """Round 6's scalar-register diet of the persistent folded kernels (docs/EXPERIMENTS.md R6.1: the parameter block re-read per
    problem through s_load, offsets one problem ahead, per-problem base pointers): v_readlane + v_writelane 926 -> 458 at head dim
    128 and 1,440 -> 574 with the research-path bias.  A change that lets hipcc keep the parameter block alive across the problem
    loop again shows up here, without a GPU.  (The verdict's <= 64 was not reached: what is left is the pair's mask / bias context
    and the DMA plans, and the headline moved by 1.1 %.)"""
    census = _lane_spill_census(("hstu_attn_bwd_fold_kernelIDF16bLi128ELi128E", "hstu_attn_bwd_fold_bias_kernelIDF16bLi64E"))
    fold = [v for k, v in census.items() if "fold_kernelIDF16bLi128ELi128E" in k]
    bias = [v for k, v in census.items() if "fold_bias_kernelIDF16bLi64E" in k]
    assert len(fold) == 1 and len(bias) == 1, census
    assert sum(fold[0]) <= 500, fold         # 458 at the commit that introduced the test
    # (574 at the commit that introduced the test; 636 since the kernel takes its users from a counter and stages the weight tables once per
    # workgroup -- three more wave-uniform values across the user loop -- which measured 16 % FASTER on the ML-20M shape: DESIGN 4.2e)
    assert sum(bias[0]) <= 680, bias
