#!/usr/bin/env python3
"""Cycle trace of one workgroup of the forward kernel at the metric shape (tools/trace_build.sh first):
python tools/trace_fwd.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generative_recommenders_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "tests", "probe", "libhstu_trace.so")
from generative_recommenders_amd.ops import _launch
dev = "cuda"
N, H, d, B = 200, 4, 128, 8192
off = torch.arange(B + 1, device=dev, dtype=torch.int64) * N
fused = torch.empty(B * N, H, 3 * d, device=dev, dtype=torch.bfloat16).uniform_(-0.01, 0.01)
q, k, v = torch.split(fused, [d, d, d], dim=-1)
trace = torch.zeros(8 * 256, dtype=torch.int64, device=dev)
L.lib()
tl = ctypes.CDLL(L.LIB_PATH)
tl.hstu_trace_set_fwd.argtypes = [ctypes.c_void_p]
tl.hstu_trace_set_fwd(trace.data_ptr())
for _ in range(3):
    _launch.attn_fwd(q, k, v, off, None, N, d ** -0.5, 1.0 / N)
torch.cuda.synchronize()
t = trace.cpu().view(8, 128, 2).numpy()
t0 = min(int(t[w, 0, 1]) for w in range(4) if t[w, 0, 0])
for w in range(4):
    print("--- wave", w)
    prev = t0
    for i in range(128):
        tag, ts = int(t[w, i, 0]), int(t[w, i, 1])
        if tag == 0:
            break
        print(f"  tag {tag:3d} t={ts - t0:8d} (+{ts - prev})")
        prev = ts
