#!/usr/bin/env python3
"""Weight-gradient GEMMs x^T dy of the two projections: slab count of the batched split (ops/mm.py::weight_grad_mm),
operand order (dW vs dW^T) and partial-sum dtype, at the layer shape and at the ops-bench shape.
python tools/sweep_wgrad.py > gpurun_out/r2/sweep_wgrad.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = "cuda"
res = {}
def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for L in (194_000, 1_550_000):
    for K, N in ((512, 2048), (1536, 512)):
        x = torch.randn(L, K, device=dev, dtype=torch.bfloat16); dy = torch.randn(L, N, device=dev, dtype=torch.bfloat16)
        flops = 2.0 * L * K * N
        row = {}
        for S in (1, 4, 8, 16, 32, 64, 128, 256):
            slab = (L // S) // 64 * 64
            main = slab * S
            xa, da = x[:main].view(S, slab, K), dy[:main].view(S, slab, N)
            for name, fn in ((f"S{S}_xTdy_f32", lambda: torch.bmm(xa.transpose(1, 2), da, out_dtype=torch.float32).sum(0)),
                             (f"S{S}_dyTx_f32", lambda: torch.bmm(da.transpose(1, 2), xa, out_dtype=torch.float32).sum(0)),
                             (f"S{S}_xTdy_bf16", lambda: torch.bmm(xa.transpose(1, 2), da).sum(0))):
                try:
                    ms = timeit(fn)
                    row[name] = dict(ms=round(ms, 3), tflops=round(flops / ms / 1e9, 1))
                except Exception as e:
                    row[name] = str(e)[:80]
        res[f"L{L}_K{K}_N{N}"] = row
        del x, dy
print(json.dumps(res, indent=1))
