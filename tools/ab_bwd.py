#!/usr/bin/env python3
"""A/B of attention-kernel variants in ONE process (one torch import, one set of inputs):

    python tools/ab_bwd.py [--workload M-full|M-jag] [--head-dim 128] [--fwd] base.so var1.so var2.so ...

Every library (tools/build_variant.sh) is loaded with ctypes next to the others and called through the C ABI
(hstu_attn_bwd, or hstu_attn_fwd with --fwd) on the metric shape; prints the HIP-event time per launch
(median of --reps rounds, the variants interleaved round-robin so that clock drift hits them alike) and
whether dq/dk/dv are bit-identical to the first library's.
"""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from generative_recommenders_amd import _lib as L  # noqa: E402
from generative_recommenders_amd.ops import _launch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--workload", default="M-full")
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--users", type=int, default=8192)
    ap.add_argument("--max-seq-len", type=int, default=200)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--fwd", action="store_true")
    ap.add_argument("--targets", action="store_true", help="num_targets = randint(1, 21) per user (the M-targets workload of bench.py)")
    a = ap.parse_args()
    dev = "cuda"
    H, d, B, N = a.heads, a.head_dim, a.users, a.max_seq_len
    gen = torch.Generator(device=dev).manual_seed(1001)
    if a.workload == "M-full":
        lengths = torch.full((B,), N, dtype=torch.int64, device=dev)
    else:
        lengths = torch.randint(int(0.9 * N), N, (B,), generator=gen, device=dev, dtype=torch.int64)
    off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(lengths, 0)
    Lt = int(off[-1])
    fused = torch.empty(Lt, H, 3 * d, device=dev, dtype=torch.bfloat16).uniform_(-0.01, 0.01, generator=gen)
    q, k, v = torch.split(fused, [d, d, d], dim=-1)
    do = torch.randn(Lt, H, d, device=dev, dtype=torch.bfloat16, generator=gen)
    out = torch.empty(Lt, H, d, device=dev, dtype=torch.bfloat16)
    dfused = torch.zeros_like(fused)
    dq, dk, dv = torch.split(dfused, [d, d, d], dim=-1)
    bp = L.HstuAttnBwdParams()
    nt = torch.randint(1, 21, (B,), generator=gen, device=dev, dtype=torch.int64) if a.targets else None
    _launch._fill_attn_params(bp.fwd, q, k, v, None, off, nt, N, d ** -0.5, 1.0 / N, 0, 0, 0, 0)
    bp.fwd.out = out.data_ptr()
    bp.fwd.o_row_stride, bp.fwd.o_head_stride = out.stride(0), out.stride(1)
    bp.dout, bp.dq, bp.dk, bp.dv = do.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    bp.do_row_stride, bp.do_head_stride = do.stride(0), do.stride(1)
    for n, t in (("dq", dq), ("dk", dk), ("dv", dv)):
        setattr(bp, n + "_row_stride", t.stride(0))
        setattr(bp, n + "_head_stride", t.stride(1))
    bp.total_rows = Lt
    st = torch.cuda.current_stream().cuda_stream
    handles = []
    for path in a.libs:
        h = C.CDLL(os.path.abspath(path))
        h.hstu_attn_bwd.argtypes = [C.POINTER(L.HstuAttnBwdParams), C.c_void_p]
        h.hstu_attn_fwd.argtypes = [C.POINTER(L.HstuAttnParams), C.c_void_p]
        h.hstu_last_error.restype = C.c_char_p
        handles.append(h)

    def call(h):
        rc = h.hstu_attn_fwd(C.byref(bp.fwd), st) if a.fwd else h.hstu_attn_bwd(C.byref(bp), st)
        if rc != 0:
            raise RuntimeError(h.hstu_last_error().decode())

    ref = None
    same = []
    for h in handles:
        dfused.zero_()
        out.zero_()
        for _ in range(2):
            call(h)
        torch.cuda.synchronize()
        got = out.clone() if a.fwd else dfused.clone()
        if ref is None:
            ref = got
        rel = float((got.float() - ref.float()).norm() / ref.float().norm())
        same.append((bool(torch.equal(got, ref)) and bool(torch.isfinite(got.float()).all()), rel))
    times = [[] for _ in handles]
    for _ in range(a.reps):
        for i, h in enumerate(handles):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.launches):
                call(h)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / a.launches)
    per_tok = H * (4 * d if a.fwd else 7 * d) * 2
    for path, t, s in zip(a.libs, times, same):
        ms = statistics.median(t)
        print(f"{os.path.basename(path):28s} {'fwd' if a.fwd else 'bwd'} {ms:7.3f} ms (min {min(t):.3f})  "
              f"{Lt * per_tok / ms / 1e6:7.0f} GB/s  frac {Lt * per_tok / ms / 1e6 / 8000:.3f}  bit-identical to first: {s[0]} (rel diff {s[1]:.2e})")


if __name__ == "__main__":
    main()
