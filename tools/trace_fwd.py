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
names = {1: "start", 2: "Q frags in regs", 3: "first tiles issued", 10: "barrier passed+dma issued", 11: "S done", 12: "silu done",
         14: "tile done", 20: "loop done", 21: "end"}
for blk, rows in (("heavy query block (rows 128..199, 7 key tiles)", range(0, 4)), ("light query block (rows 0..127, 4 key tiles)", range(4, 8))):
    live = [w for w in rows if t[w, 0, 0]]
    if not live:
        continue
    t0 = min(int(t[w, 0, 1]) for w in live)
    occ = {w: {} for w in live}
    table = {}
    for w in live:
        for i in range(128):
            tag, ts = int(t[w, i, 0]), int(t[w, i, 1])
            if tag == 0:
                break
            k = occ[w].get(tag, 0)
            occ[w][tag] = k + 1
            table.setdefault((tag, k), {})[w] = ts - t0
    print(f"===== FORWARD {blk}: cycles since the first wave's start")
    print(f"{'mark':>30s} " + " ".join(f"{'w' + str(w % 4):>7s}" for w in live))
    for (tag, k), d in sorted(table.items(), key=lambda kv: min(kv[1].values())):
        print(f"{names.get(tag, str(tag)):>27s}#{k} " + " ".join(f"{d[w]:7d}" if w in d else "      ." for w in live))
