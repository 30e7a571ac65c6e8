#!/usr/bin/env python3
"""Lint of the wide backward's code object (the kernel issues its MFMAs through inline asm, which hipcc neither schedules
around nor pads: csrc/hstu_attn_bwd_wide.cuh).  Reads the assembly hipcc emits for a translation unit and checks, for every
kernel whose name contains `--kernel` (default: hstu_attn_bwd_wide):

  1. no register spilled, no scratch;
  2. the compiler itself never touches the accumulator file (no v_accvgpr_* and no a[...] operand outside ;;#ASMSTART .. ;;#ASMEND):
     all 256 AGPRs belong to the asm statements;
  3. MFMA write -> VALU / memory read-or-write hazard: after an asm MFMA that writes VGPRs, no non-MFMA instruction touches those
     registers within 12 wait states (8-pass XDL: 11) -- the chains end in an `s_nop 15` drain, this checks that the compiler
     put nothing of its own (a copy, a spill) in between;
  4. VALU write -> MFMA read: an asm MFMA is preceded by its own `s_nop 1` (2 wait states) inside the statement.

    python tools/lint_asm_mfma.py [file.s]      (without a file: compiles csrc/attn_wide_bf16.hip and attn_wide_f16.hip)
Exit status 1 on a finding.  tests/test_kernel_resources.py runs it."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "generative_recommenders_amd", "csrc")


def regs_of(tok, kind):
    """set of register indices of class `kind` ('v' | 'a') named in an operand string"""
    out = set()
    for m in re.finditer(r"\b%s\[(\d+):(\d+)\]" % kind, tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\b%s(\d+)\b" % kind, tok):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\b%s\[(\d+)\]" % kind, tok):
        out.add(int(m.group(1)))
    return out


def lint(path, kernel_pat):
    findings = []
    lines = open(path).read().splitlines()
    in_kernel = False
    in_asm = False
    name = None
    hot = {}      # vgpr index -> remaining wait states
    n_mfma = 0
    prev_in_asm_nop = 0
    for ln, raw in enumerate(lines, 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            name = m.group(1)
            in_kernel = kernel_pat in name
            hot = {}
            continue
        if not in_kernel:
            if raw.strip().startswith(".vgpr_spill_count:") or raw.strip().startswith(".private_segment_fixed_size:"):
                pass
            continue
        if raw.strip().startswith(";;#ASMSTART"):
            in_asm = True
            prev_in_asm_nop = 0
            continue
        if raw.strip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        if raw.strip().startswith("s_endpgm"):
            in_kernel = False
            continue
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        op = line.split()[0]
        rest = line[len(op):]
        states = 1
        if op == "s_nop":
            states = int(rest.strip()) + 1
        if op.startswith("v_mfma"):
            n_mfma += 1
            if not in_asm:
                findings.append(f"{path}:{ln}: MFMA outside an asm statement in {name}: {line}")
            else:
                if prev_in_asm_nop < 2:
                    findings.append(f"{path}:{ln}: asm MFMA without its 2 wait states in front: {line}")
                ops = [t.strip() for t in rest.split(",")]
                dst = regs_of(ops[0], "v")
                # an MFMA may read / write hot registers (accumulate chain, pipe interlocks its own issue)
                for r in list(hot):
                    hot[r] -= 8
                    if hot[r] <= 0:
                        del hot[r]
                for r in dst:
                    hot[r] = 12
            prev_in_asm_nop = 0
            continue
        if in_asm and op == "s_nop":
            prev_in_asm_nop += states
        # rule 2
        if not in_asm and (op.startswith("v_accvgpr") or regs_of(rest, "a")):
            findings.append(f"{path}:{ln}: compiler instruction touches the accumulator file in {name}: {line}")
        # rule 3
        if hot:
            touched = regs_of(rest, "v") & set(hot)
            if touched and not op.startswith("s_nop"):
                findings.append(f"{path}:{ln}: {op} touches v{sorted(touched)[0]}.. {min(hot[r] for r in touched)} wait states after an asm MFMA wrote it: {line}")
            for r in list(hot):
                hot[r] -= states
                if hot[r] <= 0:
                    del hot[r]
    # rule 1: metadata
    text = "\n".join(lines)
    for blk in text.split("- .agpr_count:")[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm or kernel_pat not in nm.group(1):
            continue
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        if spill or scratch:
            findings.append(f"{path}: {nm.group(1)}: {spill} spilled registers, {scratch} bytes of scratch")
    return findings, n_mfma


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--kernel", default="hstu_attn_bwd_fold")
    ap.add_argument("--flags", default="")
    a = ap.parse_args()
    files = list(a.files)
    tmp = None
    if not files:
        tmp = tempfile.mkdtemp()
        for tu in ("attn_fold_bf16", "attn_long_bf16"):      # (live translation units; --kernel hstu_attn_bwd_dkv for the second)
            out = os.path.join(tmp, tu + ".s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I.", "-I../../include", "--cuda-device-only", "-S",
                   tu + ".hip", "-o", out] + a.flags.split()
            subprocess.run(cmd, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
            files.append(out)
    bad = 0
    for f in files:
        findings, n = lint(f, a.kernel)
        print(f"{os.path.basename(f)}: {n} MFMAs checked, {len(findings)} findings")
        for x in findings[:40]:
            print("  " + x)
        bad += len(findings)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
