"""Time the fused LayerNorm + UVQK projection kernel (csrc/hstu_ln_linear.cuh) against what it replaces
(hstu_layer_norm_fwd + hipBLASLt through torch) at the layer section's shape; HIP events, one JSON line.
    python tools/bench_ln_linear.py [--rows 204800] [--n 2048] [--iters 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_recommenders_amd.ops import _launch  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=204800)
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--fused-only", action="store_true", help="time the fused kernel only (variant libraries: HSTU_HIP_LIBRARY)")
    ap.add_argument("--k512", action="store_true", help="the kernel without its LayerNorm (hstu_linear_k512) against torch.mm at the output "
                    "stage's data-gradient shape: d y = d out . W_out^T, rows x 512 -> n (default n for this mode: 1536)")
    a = ap.parse_args()
    dev, dt, k = "cuda", torch.bfloat16, 512
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(a.rows, k, device=dev, generator=g).to(dt)
    lw = (1 + 0.1 * torch.randn(k, device=dev, generator=g)).to(dt)
    lb = (0.1 * torch.randn(k, device=dev, generator=g)).to(dt)
    w_nk = (torch.randn(a.n, k, device=dev, generator=g) / k**0.5).to(dt)
    b = (0.1 * torch.randn(a.n, device=dev, generator=g)).to(dt)
    if a.k512 and a.n == 2048:
        a.n = 1536
    flops = 2.0 * a.rows * k * a.n
    if a.k512:
        w_out = (torch.randn(a.n, k, device=dev, generator=g) / k**0.5).to(dt)     # `_output_weight` as stored: (3 H d, D)
        res = {"rows": a.rows, "k": k, "n": a.n, "dtype": "bf16", "what": "d y = d out . W_out^T"}
        for _ in range(40):
            _launch.linear_k512(x, w_out)
        res["hipblaslt_mm_us"] = timed(lambda: torch.mm(x, w_out.t()), a.iters)
        res["linear_k512_us"] = timed(lambda: _launch.linear_k512(x, w_out), a.iters)
        for key in ("hipblaslt_mm_us", "linear_k512_us"):
            res[key.replace("_us", "_tflops")] = round(flops / res[key] / 1e6, 1)
        res["bit_identical"] = bool(torch.equal(_launch.linear_k512(x, w_out), torch.mm(x, w_out.t())))
        print(json.dumps({k_: (round(v, 2) if isinstance(v, float) else v) for k_, v in res.items()}))
        return

    def unfused():
        nx, _, _ = _launch.layer_norm_fwd(x, lw, lb, 1e-6)
        return torch.nn.functional.linear(nx, w_nk, b)

    res = {"rows": a.rows, "k": k, "n": a.n, "dtype": "bf16", "lib": os.path.basename(os.environ.get("HSTU_HIP_LIBRARY", "libhstu_hip.so"))}
    if a.fused_only:
        for _ in range(60):      # ~30 ms of the same kernel first: the clocks of a cold device ramp for tens of milliseconds
            _launch.ln_linear_fwd(x, lw, lb, 1e-6, w_nk, b)
        res["fused_us"] = round(timed(lambda: _launch.ln_linear_fwd(x, lw, lb, 1e-6, w_nk, b), a.iters), 2)
        res["fused_tflops"] = round(flops / res["fused_us"] / 1e6, 1)
        print(json.dumps(res))
        return
    nx, _, _ = _launch.layer_norm_fwd(x, lw, lb, 1e-6)
    res["layer_norm_us"] = timed(lambda: _launch.layer_norm_fwd(x, lw, lb, 1e-6), a.iters)
    res["hipblaslt_linear_us"] = timed(lambda: torch.nn.functional.linear(nx, w_nk, b), a.iters)
    res["unfused_us"] = timed(unfused, a.iters)
    res["fused_us"] = timed(lambda: _launch.ln_linear_fwd(x, lw, lb, 1e-6, w_nk, b), a.iters)
    res["fused_with_normed_us"] = timed(lambda: _launch.ln_linear_fwd(x, lw, lb, 1e-6, w_nk, b, want_normed=True), a.iters)
    for key in ("hipblaslt_linear_us", "unfused_us", "fused_us", "fused_with_normed_us"):
        res[key.replace("_us", "_tflops")] = round(flops / res[key] / 1e6, 1)
    y, _, _, _ = _launch.ln_linear_fwd(x, lw, lb, 1e-6, w_nk, b)
    ref = unfused()
    res["rel_fro_vs_unfused"] = float((y.float() - ref.float()).norm() / ref.float().norm())
    res["max_abs_vs_unfused"] = float((y.float() - ref.float()).abs().max())
    print(json.dumps({k_: (round(v, 2) if isinstance(v, float) and v > 1 else v) for k_, v in res.items()}))


if __name__ == "__main__":
    main()
