#!/usr/bin/env python3
"""Research-path attention (relative position + time bias fused in the kernels) at the ML-20M shape scaled to a full
GPU batch: 8192 users, N = 211, 4 heads of 64, bf16 -- against the same shape without bias (ops path)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import generative_recommenders_amd.research.modeling.sequential.hstu as R
from generative_recommenders_amd.ops.hstu_attention import hstu_mha
dev = "cuda"
torch.manual_seed(0)
# `books`: the Amazon-Books research configuration (N = 61, 4 heads of 16, long-tail lengths: randint(0, 30), 5 % at 61)
if len(sys.argv) > 1 and sys.argv[1] == "books":
    B, n, H, d = 8192, 61, 4, 16
    lengths = torch.randint(1, 30, (B,), device=dev)
    lengths[torch.rand(B, device=dev) < 0.05] = n
else:
    B, n, H, d = 8192, 211, 4, 64
    lengths = torch.randint(n // 2, n + 1, (B,), device=dev)
off = torch.zeros(B + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(lengths, 0)
L = int(off[-1])
ts = torch.sort(torch.randint(0, 10**8, (B, n), device=dev), dim=1).values
q, k, v = (torch.randn(L, H * d, device=dev, dtype=torch.bfloat16).mul_(0.3).requires_grad_() for _ in range(3))
g = torch.randn(L, H * d, device=dev, dtype=torch.bfloat16)
bias = R.RelativeBucketedTimeAndPositionBasedBias(n, 128).to(dev)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def fb_bias():
    for t in (q, k, v):
        t.grad = None
    R.hstu_rel_bias_attention(H, d, d, q, k, v, off, ts, n, bias).backward(g)


def f_bias():
    with torch.no_grad():
        R.hstu_rel_bias_attention(H, d, d, q, k, v, off, ts, n, bias)


q3, k3, v3 = (t.detach().view(L, H, d).requires_grad_() for t in (q, k, v))
g3 = g.view(L, H, d)


def fb_plain():
    for t in (q3, k3, v3):
        t.grad = None
    hstu_mha(n, d**-0.5, q3, k3, v3, off).backward(g3)


def f_plain():
    with torch.no_grad():
        hstu_mha(n, d**-0.5, q3, k3, v3, off)


es = 2
fwd_b, bwd_b = L * H * 4 * d * es, L * H * 7 * d * es
r = {"rows": L, "with_bias_fwd_ms": timed(f_bias), "with_bias_fwd_bwd_ms": timed(fb_bias), "plain_fwd_ms": timed(f_plain),
     "plain_fwd_bwd_ms": timed(fb_plain)}
r["with_bias_fwd_GBps"] = fwd_b / r["with_bias_fwd_ms"] / 1e6
r["with_bias_bwd_GBps"] = bwd_b / (r["with_bias_fwd_bwd_ms"] - r["with_bias_fwd_ms"]) / 1e6
r["plain_fwd_GBps"] = fwd_b / r["plain_fwd_ms"] / 1e6
r["plain_bwd_GBps"] = bwd_b / (r["plain_fwd_bwd_ms"] - r["plain_fwd_ms"]) / 1e6
print(json.dumps({k_: round(v_, 3) if isinstance(v_, float) else v_ for k_, v_ in r.items()}, indent=1))

if "--kernels" in sys.argv:       # device time per kernel of the with-bias forward + backward (is the op call launch-bound?)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            fb_bias()
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
    for e in rows[:12]:
        print(f"{e.self_device_time_total / 10:9.1f} us/iter  n={e.count // 10:3d}  {e.key[:110]}")
