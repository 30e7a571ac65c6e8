#!/usr/bin/env python3
"""The layer section of bench.py alone, a few steps (the command tools/prof_layer_pmc.sh profiles)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

args = argparse.Namespace(max_seq_len=200, heads=4, head_dim=128, layer_users_per_gpu=1024, layer_steps=3, layer_dropout=0.1)
print(bench.layer_section(args, 0, 1, torch.device("cuda", 0))["ms_per_step"])
