#!/usr/bin/env python3
"""Run the traced backward kernel (tests/probe/libhstu_trace.so, built by tools/trace_build.sh)
on the metric shape and print the per-wave phase timeline of one workgroup (cycles)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from generative_recommenders_amd import _lib as L
L.LIB_PATH = os.environ.get("HSTU_TRACE_LIB", os.path.join(ROOT, "tests", "probe", "libhstu_trace.so"))
from generative_recommenders_amd.ops import _launch

dev = "cuda"
H, d, B, N = 4, 128, 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 200
lengths = torch.full((B,), N, dtype=torch.int64, device=dev)
off = torch.zeros(B + 1, dtype=torch.int64, device=dev); off[1:] = torch.cumsum(lengths, 0)
Lt = int(off[-1])
fused = torch.empty(Lt, H, 3 * d, device=dev, dtype=torch.bfloat16).uniform_(-0.01, 0.01)
q, k, v = torch.split(fused, [d, d, d], dim=-1)
do = torch.randn(Lt, H, d, device=dev, dtype=torch.bfloat16)
dfused = torch.empty_like(fused); dq, dk, dv = torch.split(dfused, [d, d, d], dim=-1)
trace = torch.zeros(8 * 256, dtype=torch.int64, device=dev)
bp = L.HstuAttnBwdParams()
_launch._fill_attn_params(bp.fwd, q, k, v, None, off, None, N, d ** -0.5, 1.0 / N, 0, 0, 0, 0)
bp.dout, bp.dq, bp.dk, bp.dv = do.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
bp.do_row_stride, bp.do_head_stride = do.stride(0), do.stride(1)
for n, t in (("dq", dq), ("dk", dk), ("dv", dv)):
    setattr(bp, n + "_row_stride", t.stride(0)); setattr(bp, n + "_head_stride", t.stride(1))
bp.total_rows = Lt
bp.workspace = trace.data_ptr()
for _ in range(3):
    L.check(L.lib().hstu_attn_bwd(C.byref(bp), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
def dump(t, names, waves):
    t0 = min(int(t[w, 0, 1]) for w in range(8) if t[w, 0, 0])
    for w in waves:
        print(f"--- wave {w}")
        prev = t0
        for i in range(128):
            tag, ts = int(t[w, i, 0]), int(t[w, i, 1])
            if tag == 0:
                break
            print(f"  {str(names.get(tag, tag)):>18s}  t={ts - t0:8d}  (+{ts - prev})")
            prev = ts

print("===== BACKWARD (workgroup 4096): cycles since the first wave's start; one column per wave")
t = trace.cpu().view(8, 128, 2).numpy()
names = {1: "start", 2: "prologue dma issued", 10: "TOP barrier passed", 11: "S,dP done", 12: "elementwise done",
         13: "pairs done", 14: "pair done", 15: "barrier1 passed", 16: "dQ gemm done", 17: "dQ stored",
         18: "stage dma+copy-out issued", 19: "dq setup done", 20: "loop done", 21: "end", 22: "dump barrier passed", 23: "dV parked", 24: "final tiles parked"}
t0 = min(int(t[w, 0, 1]) for w in range(8) if t[w, 0, 0])
# waves skip marks 11/12 when they have no pair: align rows by (tag, occurrence)
rows = []
occ = [dict() for _ in range(8)]
table = {}
for w in range(8):
    for i in range(128):
        tag, ts = int(t[w, i, 0]), int(t[w, i, 1])
        if tag == 0:
            break
        k = occ[w].get(tag, 0)
        occ[w][tag] = k + 1
        table.setdefault((tag, k), {})[w] = ts - t0
order = sorted(table.items(), key=lambda kv: min(kv[1].values()))
print(f"{'mark':>28s} " + " ".join(f"{'w' + str(w):>7s}" for w in range(8)))
for (tag, k), d in order:
    print(f"{names.get(tag, str(tag)):>25s}#{k} " + " ".join(f"{d[w]:7d}" if w in d else "      ." for w in range(8)))
