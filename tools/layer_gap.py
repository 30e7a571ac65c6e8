#!/usr/bin/env python3
"""Is the layer step (bench.py's `layer` section: 3 STU layers fwd+bwd, 1024 users) GPU-bound or launch-bound?
wall time per step vs the sum of its kernels' durations (torch.profiler), and the same step replayed from a HIP graph."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from generative_recommenders_amd import data_parallel as dp
from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig, STUStack

dev = torch.device("cuda", 0)
N, H, d, B = 200, 4, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
D = H * d
gen = torch.Generator(device=dev).manual_seed(2002)
lengths = bench.make_lengths("M-jag", B, N, gen, dev)
off = dp.local_offsets(lengths)
L = int(off[-1].item())
x = torch.randn(L, D, device=dev, dtype=torch.bfloat16, generator=gen).requires_grad_()
gy = torch.randn(L, D, device=dev, dtype=torch.bfloat16, generator=gen)
nt = torch.minimum(torch.randint(1, 21, (B,), generator=gen, device=dev), lengths)
res = {}
for recompute in (True, False):
    torch.manual_seed(7)
    stack = STUStack([STULayer(STULayerConfig(embedding_dim=D, num_heads=H, hidden_dim=d, attention_dim=d, output_dropout_ratio=0.0,
                                              use_group_norm=True, recompute_normed_x=recompute, recompute_uvqk=recompute,
                                              recompute_y=recompute)) for _ in range(3)]).to(dev)

    def step():
        for p in stack.parameters():
            p.grad = None
        x.grad = None
        y = stack(x=x, x_lengths=lengths, x_offsets=off, max_seq_len=N, num_targets=nt)
        y.backward(gy)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10 * 1e3
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ksum = sum(e.device_time for e in ev) / 5 / 1e3
    top = {}
    for e in ev:
        top[e.name[:60]] = top.get(e.name[:60], 0) + e.device_time / 5 / 1e3
    res["recompute" if recompute else "keep"] = dict(wall_ms=round(wall, 3), kernel_sum_ms=round(ksum, 3), kernels_per_step=len(ev) // 5,
                                                     top={k: round(v, 3) for k, v in sorted(top.items(), key=lambda kv: -kv[1])[:14]})
print(json.dumps(res, indent=1))
