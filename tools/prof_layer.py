#!/usr/bin/env python3
"""Layer-path kernel breakdown (3 STU layers fwd+bwd, as bench.py's `layer` section) with torch.profiler:
prints the top device kernels by total time.  Run on the GPU box: python tools/prof_layer.py"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

args = argparse.Namespace(max_seq_len=200, heads=4, head_dim=128, layer_users_per_gpu=1024, layer_steps=5, layer_dropout=0.1)
dev = torch.device("cuda", 0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    r = bench.layer_section(args, 0, 1, dev)
print(r)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
# device kernels only, all of them, per step
steps = 24.0   # (layer_section: 5 timed + warmup, twice more for dropout_off / no_recompute -- see the '# of Calls' column)
rows = [e for e in prof.key_averages() if getattr(e, "device_type", None) is not None and e.self_device_time_total > 0 and not e.key.startswith(("aten::", "autograd::", "##", "_"))  or e.key.startswith("_ZN")]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"device kernels: {len(rows)} names, {tot / 1e3:.1f} ms total")
for e in rows[:70]:
    print(f"{e.self_device_time_total / 1e3:9.2f} ms {100 * e.self_device_time_total / tot:5.1f}%  n={e.count:5d}  avg={e.self_device_time_total / max(e.count, 1):8.1f} us  {e.key[:150]}")
