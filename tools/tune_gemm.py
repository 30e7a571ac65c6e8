#!/usr/bin/env python3
"""The six projection GEMMs of one STU layer with PyTorch's TunableOp (hipBLASLt / rocBLAS solution search per shape) off
and on.  python tools/tune_gemm.py [rows]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd.ops.mm import weight_grad_mm

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 194_000
D = 512
dev = "cuda"
dt = torch.bfloat16
x = torch.randn(rows, D, device=dev, dtype=dt)
w_uvqk = torch.randn(D, 4 * D, device=dev, dtype=dt)
b_uvqk = torch.randn(4 * D, device=dev, dtype=dt)
g_uvqk = torch.randn(rows, 4 * D, device=dev, dtype=dt)
y3 = torch.randn(rows, 3 * D, device=dev, dtype=dt)
w_out = torch.randn(3 * D, D, device=dev, dtype=dt)
g_out = torch.randn(rows, D, device=dev, dtype=dt)
cases = {
    "uvqk_fwd": (lambda: torch.addmm(b_uvqk, x, w_uvqk), 2.0 * rows * D * 4 * D),
    "uvqk_dgrad": (lambda: torch.mm(g_uvqk, w_uvqk.t()), 2.0 * rows * D * 4 * D),
    "uvqk_wgrad": (lambda: weight_grad_mm(x, g_uvqk), 2.0 * rows * D * 4 * D),
    "out_fwd": (lambda: torch.addmm(x, y3, w_out), 2.0 * rows * 3 * D * D),
    "out_dgrad": (lambda: torch.mm(g_out, w_out.t()), 2.0 * rows * 3 * D * D),
    "out_wgrad": (lambda: weight_grad_mm(y3, g_out), 2.0 * rows * 3 * D * D),
}
res = {}
for mode in ("off", "on"):
    if mode == "on":
        import torch.cuda.tunable as tn
        tn.enable(True)
        tn.tuning_enable(True)
        tn.set_max_tuning_duration(30)
        tn.set_max_tuning_iterations(20)
        try:
            tn.write_file_on_exit(False)
        except Exception:
            pass
    for name, (fn, flops) in cases.items():
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.setdefault(name, {})[mode] = dict(ms=round(ms, 4), tflops=round(flops / ms / 1e9, 1), first_calls_s=round(first, 2))
print(json.dumps(res, indent=1))
tot = {m: sum(v[m]["ms"] for v in res.values()) for m in ("off", "on")}
print("all six, ms:", tot)
