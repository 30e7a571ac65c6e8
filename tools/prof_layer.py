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
