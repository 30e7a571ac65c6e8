#!/usr/bin/env python3
"""hipBLASLt throughput of the projection shapes by operand layout (is a transposed weight copy worth keeping?)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = "cuda"; bf = torch.bfloat16
L, D = 194560, 512


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


out = {}
for name, K, N in (("uvqk", D, 4 * D), ("output", 3 * D, D)):
    x = torch.randn(L, K, device=dev, dtype=bf)
    W = torch.randn(K, N, device=dev, dtype=bf) * 0.02          # reference layout (in, out)
    Wt = W.t().contiguous()                                       # (out, in)
    b = torch.zeros(N, device=dev, dtype=bf)
    g = torch.randn(L, N, device=dev, dtype=bf)
    fl = 2.0 * L * K * N
    r = {}
    r["fwd NN  x @ W"] = fl / timed(lambda: torch.mm(x, W)) / 1e12
    r["fwd NT  x @ Wt.t()"] = fl / timed(lambda: torch.mm(x, Wt.t())) / 1e12
    r["fwd addmm NN"] = fl / timed(lambda: torch.addmm(b, x, W)) / 1e12
    r["fwd linear(x, Wt, b)"] = fl / timed(lambda: torch.nn.functional.linear(x, Wt, b)) / 1e12
    r["dgrad NT g @ W.t()"] = fl / timed(lambda: torch.mm(g, W.t())) / 1e12
    r["dgrad NN g @ Wt"] = fl / timed(lambda: torch.mm(g, Wt)) / 1e12
    r["wgrad TN x.t() @ g"] = fl / timed(lambda: torch.mm(x.t(), g)) / 1e12
    r["wgrad (g.t() @ x) -> Wt grad"] = fl / timed(lambda: torch.mm(g.t(), x)) / 1e12
    out[name] = {k: round(v, 1) for k, v in r.items()}
print(json.dumps(out, indent=1))
