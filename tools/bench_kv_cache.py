#!/usr/bin/env python3
"""M-FALCON microbatch step of one STU layer at the HSTU-large shape (BASELINE config 5: D = 1024, 16 heads of 64,
32 users with ~8K cached history rows, 256 candidate rows per user per microbatch): ``cached_forward`` with the
in-place KV append (no_grad: only the delta rows are written) vs the reference's rebuild of [cache ; delta] by
concat_2D_jagged on every call (modules/stu.py:134-172).  Prints one JSON object.
Run on the GPU box:  python tools/bench_kv_cache.py > gpurun_out/kv_cache.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig
from generative_recommenders_amd.ops.jagged_tensors import asynchronous_complete_cumsum

dev = "cuda"
torch.manual_seed(0)
D, H, d, B, N, delta = 1024, 16, 64, 32, 8192, 256
layer = STULayer(STULayerConfig(embedding_dim=D, num_heads=H, hidden_dim=d, attention_dim=d, output_dropout_ratio=0.0,
                                target_aware=True, use_group_norm=False), is_inference=True).to(dev).eval()
lengths = torch.randint(int(0.9 * N), N - delta, (B,), device=dev)
off = asynchronous_complete_cumsum(lengths)
L = int(off[-1])
x = torch.randn(L, D, device=dev, dtype=torch.bfloat16) * 0.1
with torch.no_grad():
    layer(x=x, x_lengths=lengths, x_offsets=off, max_seq_len=N, num_targets=torch.zeros_like(lengths),
          max_kv_caching_len=N, kv_caching_lengths=lengths)
dx = torch.randn(B * delta, D, device=dev, dtype=torch.bfloat16) * 0.1
nt = torch.full((B,), delta, device=dev, dtype=lengths.dtype)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def step_in_place():
    with torch.no_grad():
        return layer.cached_forward(delta_x=dx, num_targets=nt)


def step_rebuild():
    with torch.enable_grad():
        return layer.cached_forward(delta_x=dx, num_targets=nt)


a = step_in_place().clone()
b = step_rebuild().detach()
t_fast, t_slow = timed(step_in_place), timed(step_rebuild)
row = D * 2                                    # bytes of one K (or V) row, all heads
hist = 2 * 2 * L * row                         # rebuild: read + write the cached K and V rows
app = 2 * 2 * B * delta * row                  # append: read + write the delta K and V rows
attn = 2 * (L + B * delta) * row + 2 * B * delta * row
print(json.dumps({
    "shape": {"D": D, "heads": H, "head_dim": d, "users": B, "cached_rows": L, "delta_rows_per_user": delta, "dtype": "bf16"},
    "identical_outputs": bool(torch.equal(a, b)),
    "cached_forward_ms": {"in_place_append": round(t_fast, 3), "rebuild_by_concat (reference algorithm)": round(t_slow, 3)},
    "speedup": round(t_slow / t_fast, 2),
    "bytes": {"kv_copy_rebuild": hist + app, "kv_copy_append": app, "delta_attention_algorithmic": attn},
}, indent=1))
