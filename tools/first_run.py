#!/usr/bin/env python3
"""Why is the forward sometimes 2.2 ms instead of 1.37 ms in the FIRST process on a fresh box?  bench.py's loop (forward and
backward alternating, HIP events around each), per-10-step averages; then the inputs are freed, re-allocated and measured
again in the same process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd.ops import _launch

dev = "cuda"
B, N, H, d = 8192, 200, 4, 128
off = torch.arange(B + 1, device=dev, dtype=torch.int64) * N


def run(tag, pad_mb=0, host_sleep=0.0):
    pad = torch.empty(pad_mb << 20, dtype=torch.uint8, device=dev) if pad_mb else None
    fused = torch.randn(B * N, H, 4 * d, device=dev, dtype=torch.bfloat16)
    q, k, v, dout = fused[..., :d], fused[..., d:2 * d], fused[..., 2 * d:3 * d], fused[..., 3 * d:]
    dfused = torch.empty(B * N, H, 3 * d, device=dev, dtype=torch.bfloat16)
    dq, dk, dv = dfused[..., :d], dfused[..., d:2 * d], dfused[..., 2 * d:]
    res = []
    for blk in range(6):
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(10)]
        t0 = time.perf_counter()
        for i in range(10):
            ev[i][0].record()
            out = _launch.attn_fwd(q, k, v, off, None, N, d**-0.5, 1.0 / N)
            ev[i][1].record()
            _launch.attn_bwd(dout, q, k, v, off, None, N, d**-0.5, 1.0 / N, dq=dq, dk=dk, dv=dv)
            ev[i][2].record()
        host = (time.perf_counter() - t0) / 10 * 1e3
        torch.cuda.synchronize()
        f = sum(e[0].elapsed_time(e[1]) for e in ev) / 10
        b = sum(e[1].elapsed_time(e[2]) for e in ev) / 10
        res.append((round(f, 3), round(b, 3), round(host, 3)))
    print(tag, "fused %x dfused %x out %x" % (fused.data_ptr(), dfused.data_ptr(), out.data_ptr()), res, flush=True)


run("first")
run("again")
torch.cuda.empty_cache()
run("after empty_cache")
run("padded 64 MB", 64)
