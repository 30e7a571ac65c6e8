#!/usr/bin/env python3
"""Timeline of one workgroup of the fused LayerNorm + projection kernel (variant library built with -DLNL_TRACE:
tools/build_variant.sh lnl_trace -DLNL_TRACE ln_linear): s_memtime stamps of lane 0 of every wave of workgroup 100."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generative_recommenders_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "tests", "probe", f"libhstu_{sys.argv[1] if len(sys.argv) > 1 else 'lnl_trace'}.so")
from generative_recommenders_amd.ops import _launch

dev, dt, k, n, rows = "cuda", torch.bfloat16, 512, 2048, 204800
x = torch.randn(rows, k, device=dev).to(dt)
lw, lb = torch.ones(k, device=dev, dtype=dt), torch.zeros(k, device=dev, dtype=dt)
w_nk = (torch.randn(n, k, device=dev) / k**0.5).to(dt)
b = torch.zeros(n, device=dev, dtype=dt)
y = torch.empty(rows, n, device=dev, dtype=dt)
mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
trace = torch.zeros(8 * 256, dtype=torch.int64, device=dev)
for _ in range(30):     # clocks up
    _launch.ln_linear_fwd(x, lw, lb, 1e-6, w_nk, b)
torch.cuda.synchronize()
for _ in range(3):
    L.check(L.lib().hstu_ln_linear_fwd(x.data_ptr(), k, lw.data_ptr(), lb.data_ptr(), 1e-6, w_nk.data_ptr(), b.data_ptr(), y.data_ptr(), n,
                                       trace.data_ptr(), k, mean.data_ptr(), rstd.data_ptr(), rows, k, n, 0, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
t = trace.cpu().view(8, 128, 2).numpy()
names = {**{20 + i: f"MFMA {4 * i + 4} issued" for i in range(8)}, 1: "start", 2: "rows: load", 3: "rows: normalised", 10: "step top", 11: "16 MFMAs issued", 12: "barrier passed"}
t0 = min(int(t[w, 0, 1]) for w in range(8))
for w in (0, 1, 4):
    print(f"--- wave {w}")
    prev = t0
    for i in range(126):
        tag, ts = int(t[w, i, 0]), int(t[w, i, 1])
        if tag == 0:
            break
        print(f"  {names.get(tag, tag):>18s}  t={ts - t0:8d}  (+{ts - prev})")
        prev = ts
# per-step summary over all waves: step top -> 16 issued -> barrier passed -> next top
import numpy as np
for w in range(8):
    tags, ts = t[w, :, 0], t[w, :, 1]
    tops = [i for i in range(126) if tags[i] == 10]
    a = [int(ts[i + 1] - ts[i]) for i in tops if i + 3 < 126 and tags[i + 1] == 11 and tags[i + 2] == 12 and tags[i + 3] == 10]
    bw = [int(ts[i + 2] - ts[i + 1]) for i in tops if i + 3 < 126 and tags[i + 1] == 11 and tags[i + 2] == 12 and tags[i + 3] == 10]
    c = [int(ts[i + 3] - ts[i + 2]) for i in tops if i + 3 < 126 and tags[i + 1] == 11 and tags[i + 2] == 12 and tags[i + 3] == 10]
    if a:
        print(f"wave {w}: first half {np.mean(a):7.0f}  wait+barrier {np.mean(bw):7.0f}  second half {np.mean(c):7.0f}  (cycles, mean of {len(a)} steps)")
