#!/usr/bin/env python3
"""One training step of bench.py's 3-layer STU stack under torch.profiler: every GPU kernel of the step with its call count and
total time, and for the glue kernels (copies, element-wise, reductions) the operator that launched them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from generative_recommenders_amd import data_parallel as dp
from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig, STUStack
import bench

dev = torch.device("cuda", 0)
N, H, d, B = 200, 4, 128, 1024
D = H * d
gen = torch.Generator(device=dev).manual_seed(2002)
lengths = bench.make_lengths("M-jag", B, N, gen, dev)
off = dp.local_offsets(lengths)
L = int(off[-1])
x = torch.randn(L, D, device=dev, dtype=torch.bfloat16, generator=gen).requires_grad_()
gy = torch.randn(L, D, device=dev, dtype=torch.bfloat16, generator=gen)
nt = torch.minimum(torch.randint(1, 21, (B,), generator=gen, device=dev), lengths)
torch.manual_seed(7)
stack = STUStack([STULayer(STULayerConfig(embedding_dim=D, num_heads=H, hidden_dim=d, attention_dim=d, output_dropout_ratio=0.1,
                                          use_group_norm=True)) for _ in range(3)]).to(dev)
stack.train()
reducer = dp.GradientAllReducer(None, buckets=[layer.parameters() for layer in stack._stu_layers], overlap=True)


def step():
    for p in stack.parameters():
        p.grad = None
    x.grad = None
    y = stack(x=x, x_lengths=lengths, x_offsets=off, max_seq_len=N, num_targets=nt)
    y.backward(gy)
    reducer.reduce()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = []
for e in ka:
    t = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
    if t > 0:
        rows.append((t, e.count, e.key[:70], str(e.input_shapes)[:90]))
rows.sort(reverse=True)
tot = 0
for t, c, k, sh in rows[:60]:
    print(f"{t:10.1f} us  x{c:<3d} {k:70s} {sh}")
