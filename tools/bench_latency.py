#!/usr/bin/env python3
"""Host-side cost per call of the hot-path entry points on a tiny problem (1 user, 64 rows): what a serving loop pays
per launch on top of the kernel (ctypes marshalling, argument checks, output allocation)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd.ops import _launch
from generative_recommenders_amd.ops.hstu_attention import hstu_mha
from generative_recommenders_amd.ops.layer_norm import layer_norm

dev = "cuda"
L, H, d = 64, 4, 128
q = torch.randn(L, H, d, device=dev, dtype=torch.bfloat16)
off = torch.tensor([0, L], device=dev)
x = torch.randn(L, H * d, device=dev, dtype=torch.bfloat16)
w = torch.ones(H * d, device=dev, dtype=torch.bfloat16)


def per_call(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


with torch.no_grad():
    out = {
        "attn_fwd (C-ABI wrapper)": per_call(lambda: _launch.attn_fwd(q, q, q, off, None, L, 0.1, 1.0 / L)),
        "hstu_mha (autograd function, no_grad)": per_call(lambda: hstu_mha(L, 0.1, q, q, q, off)),
        "layer_norm": per_call(lambda: layer_norm(x, w, w, 1e-6)),
        "complete_cumsum": per_call(lambda: _launch.complete_cumsum(off[1:])),
        "torch baseline: x + x": per_call(lambda: x + x),
        "torch baseline: F.layer_norm": per_call(lambda: torch.nn.functional.layer_norm(x, (H * d,), w, w, 1e-6)),
    }
print(json.dumps({k: round(v, 1) for k, v in out.items()} | {"unit": "us per call, back to back, 1 user x 64 rows"}, indent=1))
