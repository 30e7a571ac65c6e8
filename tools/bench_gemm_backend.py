#!/usr/bin/env python3
"""The layer's six projection calls under torch's two BLAS back ends on ROCm (hipBLASLt = default here, rocBLAS =
preferred_blas_library("cublas")): TFLOP/s per call at the layer section's shape."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd.ops.mm import weight_grad_mm
dev = "cuda"; bf = torch.bfloat16
L, D = 196000, 512


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


x = torch.randn(L, D, device=dev, dtype=bf)
W = torch.randn(D, 4 * D, device=dev, dtype=bf) * 0.02
Wt = W.t().contiguous()
b = torch.zeros(4 * D, device=dev, dtype=bf)
g = torch.randn(L, 4 * D, device=dev, dtype=bf)
y3 = torch.randn(L, 3 * D, device=dev, dtype=bf)
Wo = torch.randn(3 * D, D, device=dev, dtype=bf) * 0.02
go = torch.randn(L, D, device=dev, dtype=bf)
cases = {
    "uvqk fwd linear(x, Wt, b)": (lambda: torch.nn.functional.linear(x, Wt, b), 2.0 * L * D * 4 * D),
    "uvqk fwd addmm(b, x, W)": (lambda: torch.addmm(b, x, W), 2.0 * L * D * 4 * D),
    "uvqk dgrad mm(g, W.t())": (lambda: torch.mm(g, W.t()), 2.0 * L * D * 4 * D),
    "uvqk wgrad (16 slabs)": (lambda: weight_grad_mm(x, g), 2.0 * L * D * 4 * D),
    "out fwd addmm(x, y3, Wo)": (lambda: torch.addmm(x, y3, Wo), 2.0 * L * 3 * D * D),
    "out dgrad mm(go, Wo.t())": (lambda: torch.mm(go, Wo.t()), 2.0 * L * 3 * D * D),
    "out wgrad (16 slabs)": (lambda: weight_grad_mm(y3, go), 2.0 * L * 3 * D * D),
}
out = {}
for rnd, lib in enumerate(("hipblaslt", "cublas", "hipblaslt", "cublas", "hipblaslt", "cublas")):      # (the first round warms the GPU up)
    torch.backends.cuda.preferred_blas_library(lib)
    r = {k: round(fl / timed(fn) / 1e12, 1) for k, (fn, fl) in cases.items()}
    if rnd >= 2:
        out.setdefault(lib, []).append(r)
print(json.dumps({lib: {k: [r[k] for r in rs] for k in cases} for lib, rs in out.items()}, indent=1))
