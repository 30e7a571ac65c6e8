#!/usr/bin/env python3
"""A/B of row-pass kernel variants (norm_ops.hip) in ONE process:

    python tools/ab_norm.py [--rows 204800] [--heads 4] [--head-dim 128] [--dropout 0.1] base.so var.so ...

Each library (tools/build_variant.sh NAME "-D..." norm_ops) is loaded with ctypes and called through the C ABI on the
layer shape: u * GroupNorm(attn) forward / backward (with the fused dropout), layer norm forward / backward, SiLU
forward / backward on the u slice of a (rows, 4 dim) buffer.  Prints the HIP-event time per launch (median of --reps
rounds, variants interleaved), the algorithmic HBM rate, and whether the outputs are bit-identical to the first library's.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rows", type=int, default=204800)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--launches", type=int, default=10)
    a = ap.parse_args()
    dev = "cuda"
    R, H, d = a.rows, a.heads, a.head_dim
    D = H * d
    gen = torch.Generator(device=dev).manual_seed(5)
    bf = torch.bfloat16
    attn = torch.randn(R, D, device=dev, dtype=bf, generator=gen)
    u = torch.randn(R, D, device=dev, dtype=bf, generator=gen)
    dy = torch.randn(R, D, device=dev, dtype=bf, generator=gen)
    w = torch.randn(H, device=dev, dtype=bf, generator=gen)
    b = torch.randn(H, device=dev, dtype=bf, generator=gen)
    lw = torch.randn(D, device=dev, dtype=bf, generator=gen)
    lb = torch.randn(D, device=dev, dtype=bf, generator=gen)
    uvqk = torch.randn(R, 4 * D, device=dev, dtype=bf, generator=gen)
    duvqk = torch.randn(R, 4 * D, device=dev, dtype=bf, generator=gen)
    dy3 = torch.randn(R, 3 * D, device=dev, dtype=bf, generator=gen)
    st = L.current_stream_ptr(torch.device(dev))
    code = L.torch_dtype_code(bf)
    seed = 0x1234567890ABCDEF
    base_lib = L.lib()
    ws_bytes = max(int(base_lib.hstu_norm_bwd_workspace_bytes(R, D)), 1 << 20)

    def make(lib):
        o = dict(y=torch.empty(R, D, device=dev, dtype=bf), mean=torch.empty(R, H, device=dev), rstd=torch.empty(R, H, device=dev),
                 dattn=torch.empty(R, D, device=dev, dtype=bf), du=torch.empty(R, D, device=dev, dtype=bf),
                 dw=torch.empty(H, device=dev), db=torch.empty(H, device=dev), ws=torch.empty(ws_bytes // 4, device=dev),
                 ly=torch.empty(R, D, device=dev, dtype=bf), lmean=torch.empty(R, device=dev), lrstd=torch.empty(R, device=dev),
                 ldx=torch.empty(R, D, device=dev, dtype=bf), ldw=torch.empty(D, device=dev), ldb=torch.empty(D, device=dev),
                 su=torch.empty(R, D, device=dev, dtype=bf), sd=torch.empty(R, 4 * D, device=dev, dtype=bf),
                 y3=torch.empty(R, 3 * D, device=dev, dtype=bf))
        P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        F = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_float))  # noqa: E731
        calls = {
            "norm_mul_fwd": (lambda: lib.hstu_norm_mul_dropout_fwd(P(attn), P(u), P(w), P(b), P(o["y"]), F(o["mean"]), F(o["rstd"]), R, H, d,
                                                                     C.c_float(1e-5), 1, 0, C.c_float(a.dropout), C.c_uint64(seed), code, st), 3 * R * D * 2),
            "norm_mul_bwd": (lambda: lib.hstu_norm_mul_dropout_bwd(P(dy), P(attn), P(u), P(w), P(b), F(o["mean"]), F(o["rstd"]), P(o["dattn"]),
                                                                     P(o["du"]), F(o["dw"]), F(o["db"]), F(o["ws"]), R, H, d, 1, 0,
                                                                     C.c_float(a.dropout), C.c_uint64(seed), code, st), 5 * R * D * 2),
            # the layer's own configuration: [u, attn, y] concat, u read in place from the uvqk buffer with SiLU applied on the fly
            "nm_silu_cat_fwd": (lambda: lib.hstu_norm_mul_silu_fwd(P(attn), P(uvqk), 4 * D, 1, P(w), P(b), P(o["y3"]), F(o["mean"]), F(o["rstd"]), R, H, d,
                                                                     C.c_float(1e-5), 1, 1, C.c_float(a.dropout), C.c_uint64(seed), code, st), 5 * R * D * 2),
            "nm_silu_cat_bwd": (lambda: lib.hstu_norm_mul_silu_bwd(P(dy3), P(attn), P(uvqk), 4 * D, 1, P(w), P(b), F(o["mean"]), F(o["rstd"]), P(o["dattn"]),
                                                                     P(o["sd"]), 4 * D, F(o["dw"]), F(o["db"]), F(o["ws"]), R, H, d, 1, 1,
                                                                     C.c_float(a.dropout), C.c_uint64(seed), code, st), 7 * R * D * 2),
            "layer_norm_fwd": (lambda: lib.hstu_layer_norm_fwd(P(attn), P(lw), P(lb), P(o["ly"]), F(o["lmean"]), F(o["lrstd"]), R, D,
                                                                 C.c_float(1e-5), code, st), 2 * R * D * 2),
            "layer_norm_bwd": (lambda: lib.hstu_layer_norm_bwd(P(dy), P(attn), P(lw), F(o["lmean"]), F(o["lrstd"]), P(o["ldx"]), F(o["ldw"]),
                                                                 F(o["ldb"]), F(o["ws"]), R, D, code, st), 3 * R * D * 2),
            "silu_fwd": (lambda: lib.hstu_silu_fwd(P(uvqk), P(o["su"]), R, D, 4 * D, D, code, st), 2 * R * D * 2),
            "silu_bwd": (lambda: lib.hstu_silu_bwd(P(duvqk), P(uvqk), P(o["sd"]), R, D, 4 * D, 4 * D, 4 * D, code, st), 3 * R * D * 2),
        }
        return o, calls

    libs = []
    for path in a.libs:
        lib = C.CDLL(os.path.abspath(path))
        for n in ("hstu_norm_mul_dropout_fwd", "hstu_norm_mul_dropout_bwd", "hstu_layer_norm_fwd", "hstu_layer_norm_bwd", "hstu_silu_fwd",
                  "hstu_silu_bwd", "hstu_norm_mul_silu_fwd", "hstu_norm_mul_silu_bwd"):
            f = getattr(lib, n)
            f.restype = C.c_int
            f.argtypes = getattr(base_lib, n).argtypes
        lib.hstu_last_error.restype = C.c_char_p
        libs.append((os.path.basename(path), lib) + make(lib))

    names = list(libs[0][3])
    times = {(ln, n): [] for ln, *_ in libs for n in names}
    for rep in range(a.reps + 1):
        for ln, lib, o, calls in libs:
            for n in names:
                fn, _ = calls[n]
                rc = fn()
                if rc:
                    raise SystemExit(f"{ln} {n}: rc={rc} {lib.hstu_last_error()}")
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.launches):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    times[(ln, n)].append(e0.elapsed_time(e1) / a.launches)
    outs_of = {"nm_silu_cat_fwd": ("y3",), "nm_silu_cat_bwd": ("dattn",), "norm_mul_fwd": ("y", "mean", "rstd"), "norm_mul_bwd": ("dattn", "du", "dw", "db"), "layer_norm_fwd": ("ly", "lmean", "lrstd"),
               "layer_norm_bwd": ("ldx", "ldw", "ldb"), "silu_fwd": ("su",), "silu_bwd": ("sd",)}
    base_o = libs[0][2]
    print(f"rows {R}, {H} heads x {d}, bf16, dropout {a.dropout}")
    for n in names:
        for ln, lib, o, calls in libs:
            ms = statistics.median(times[(ln, n)])
            same = all(torch.equal(o[k][:, :D] if k == "sd" else o[k], base_o[k][:, :D] if k == "sd" else base_o[k]) for k in outs_of[n])
            print(f"{n:16s} {ln:28s} {ms * 1e3:8.1f} us  {calls[n][1] / ms / 1e9:6.2f} TB/s  identical_to_first={same}")


if __name__ == "__main__":
    main()
