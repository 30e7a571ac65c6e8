#!/usr/bin/env python3
"""Per-(user, head) kernel cost versus sequence length: separates fixed (prologue /
epilogue) from per-tile cost.  Prints microseconds per (user, head) problem per CU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd.ops import _launch

dev = "cuda"
H, d = 4, int(os.environ.get("HD", "128"))
TOKENS = int(os.environ.get("TOKENS", str(8192 * 200)))   # users = TOKENS / N: constant work per launch
for N in [int(x) for x in (sys.argv[1:] or "32 64 96 128 160 192 200 224".split())]:
    B = max(TOKENS // N, 1)
    lengths = torch.full((B,), N, dtype=torch.int64, device=dev)
    off = torch.zeros(B + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(lengths, 0)
    L = int(off[-1])
    fused = torch.empty(L, H, 3 * d, device=dev, dtype=torch.bfloat16).uniform_(-0.01, 0.01)
    q, k, v = torch.split(fused, [d, d, d], dim=-1)
    if os.environ.get('CONTIG'):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    do = torch.randn(L, H, d, device=dev, dtype=torch.bfloat16)
    dfused = torch.empty_like(fused); dq, dk, dv = torch.split(dfused, [d, d, d], dim=-1)
    if os.environ.get('CONTIG'):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    a = d ** -0.5
    def run(n):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for _ in range(n):
            e[0].record(); o = _launch.attn_fwd(q, k, v, off, None, N, a, 1.0 / N); e[1].record()
            _launch.attn_bwd(do, q, k, v, off, None, N, a, 1.0 / N, dq=dq, dk=dk, dv=dv); e[2].record()
            torch.cuda.synchronize(); tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        return tf / n, tb / n
    run(3); tf, tb = run(10)
    nprob = B * H / 256.0
    fb, bb = L * H * 4 * d * 2, L * H * 7 * d * 2
    flops = 4.0 * B * H * N * (N + 1) / 2 * d      # causal half of QK^T and PV
    print(f"N={N:5d} users {B:6d} fwd {flops/tf/1e9:6.1f} TF/s  bwd {2.5*flops/tb/1e9:6.1f} TF/s |  fwd {tf:7.3f} ms ({tf*1e3/nprob:6.2f} us/problem/CU, {fb/tf/1e6:6.0f} GB/s)  "
          f"bwd {tb:7.3f} ms ({tb*1e3/nprob:6.2f} us/problem/CU, {bb/tb/1e6:6.0f} GB/s)", flush=True)
