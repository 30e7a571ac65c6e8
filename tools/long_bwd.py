#!/usr/bin/env python3
"""Long-sequence attention (several key blocks: general backward with fp32 dq accumulation): fwd / bwd time and TFLOP/s of
causal work.  python tools/long_bwd.py [N] [users] [heads] [d]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd.ops import _launch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4
d = int(sys.argv[4]) if len(sys.argv) > 4 else 64
dev = "cuda"
off = torch.arange(B + 1, device=dev, dtype=torch.int64) * N
L = B * N
fused = torch.randn(L, H, 4 * d, device=dev, dtype=torch.bfloat16) * 0.1
q, k, v, do = fused[..., :d], fused[..., d:2 * d], fused[..., 2 * d:3 * d], fused[..., 3 * d:]
dfused = torch.empty(L, H, 3 * d, device=dev, dtype=torch.bfloat16)
dq, dk, dv = dfused[..., :d], dfused[..., d:2 * d], dfused[..., 2 * d:]


def timed(fn, n=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


f_ms = timed(lambda: _launch.attn_fwd(q, k, v, off, None, N, d**-0.5, 1.0 / N))
b_ms = timed(lambda: _launch.attn_bwd(do, q, k, v, off, None, N, d**-0.5, 1.0 / N, dq=dq, dk=dk, dv=dv))
pairs = B * H * N * (N + 1) / 2
print(json.dumps(dict(N=N, users=B, heads=H, d=d, fwd_ms=round(f_ms, 3), bwd_ms=round(b_ms, 3),
                      fwd_tflops=round(4 * d * pairs / f_ms / 1e9, 1), bwd_tflops=round(10 * d * pairs / b_ms / 1e9, 1),
                      bwd_kernel=_launch.attn_bwd_kernel_name(torch.bfloat16, d, d, N, heads=H))))
