#!/usr/bin/env python3
"""bench.py -- HSTU attention fwd+bwd throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload M-full|M-jag|M-targets|C2|C3|C3-bias|C4|C5]

A *step* is one pass of the hot path -- hstu attention forward + backward through the C ABI (libhstu_hip.so) -- over one
batch of synthetic jagged user sequences that is already resident in HBM.  Default workload = the metric shape M
(SURVEY.md 8d): N=200, H=4, dqk=dv=128 (d=512), bf16, 8192 users per GPU; q, k, v are strided views of one fused
(sum L, H, 3*128) buffer drawn uniform(-0.01, 0.01) with seed 1001 (as ops/benchmarks/hstu_attention_bench.py:194-233
builds them); dout = randn.  The other workloads are the remaining BASELINE.json configurations as synthetic inputs
(SURVEY 8d "Other configs"): C2 research-path relative-bias attention (N=211, 4x64), C3 Amazon-Books-like long-tail
lengths (N=61, 4x16), C4 one rank's shard of the DP-8 batch (N=200, 4x64), C5 delta-q microbatches over long
histories (16x64, N<=8192, forward only, 24 layers back to back).

N > 1: one process per GPU.  Under ``torch.distributed.run`` the ranks come from the environment; called plainly with
``--gpus N`` the script spawns its N ranks itself (as the reference's main.py:68-78 does).  Users shard across ranks (weak
scaling: per-GPU work fixed); attention has no parameters, hence no collective in the timed region -- the ranks meet at
the barriers that bracket it.  The secondary "layer" section times 3 STU layers (D=512) fwd+bwd WITH the RCCL gradient
all-reduce inside the step, and the "rccl" object reports the communicator size and the bus bandwidth of the 22 MB
all-reduce that step performs.

Every section runs ~0.3 s of its own steps UNTIMED in front of its W warm-up steps (``--prewarm-s``: the clock / power ramp after an
idle gap otherwise lands in the first counted steps), then W warm-up steps, then exactly K timed steps between barrier +
synchronize.  Rank 0 prints ONE JSON line (contract in the task statement) with the extra objects
  roofline      -- dominant kernel (backward): algorithmic bytes / HIP-event kernel time vs 8 TB/s; kernel names come from
                   the library (hstu_attn_*_kernel_name); ``traffic`` = HBM bytes from committed PMC passes of the SAME
                   workload / dtype / head dim / users (``traffic_source`` says which), null otherwise
  cpu_baseline  -- the padded-dense CPU port of the reference's PyTorch path on the host cores (attention and 3 STU layers)
  parity_at_this_size -- 32 users of the timed batch against the fp64 oracle (after the timed loop)
  telemetry / calibration -- clocks, power, temperatures around the timed loops; what this box sustains for an MFMA chain, a read
                   stream and a device copy (csrc/aux_ops.hip), and the product kernels' ratios to them
"""

import argparse
import json
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before the first torch.cuda call (dmabuf-only IPC hosts: RCCL needs it)
import socket
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md); measured float4 copy 6.29 TB/s
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16

# workload -> (max_seq_len, heads, head_dim, default users per GPU, description)
WORKLOADS = {
    "M-full": (200, 4, 128, 8192, "every user has L = 200"),
    "M-jag": (200, 4, 128, 8192, "L = randint(180, 200) (generate_sparse_seq_len, sparsity 0.95)"),
    "M-targets": (200, 4, 128, 8192, "M-jag lengths, num_targets = randint(1, 21) (general mask algebra)"),
    "C2": (211, 4, 64, 8192, "ML-20M shape, research path: relative position + time-bucket bias inside the kernels, L = randint(1, 212)"),
    "C3": (61, 4, 16, 8192, "Amazon-Books shape, long-tail lengths: randint(0, 30), 5 % of the users at the full 61"),
    "C3-bias": (61, 4, 16, 8192, "Amazon-Books shape on the research path (the configuration the reference trains it with): relative position + time-bucket "
                                   "bias inside the short-sequence kernels, lengths as C3"),
    "C4": (200, 4, 64, 1024, "one rank's shard (1024 users) of the DP-8 synthetic ML-3B batch, M-jag lengths"),
    "C5": (8192, 16, 64, 32, "HSTU-large M-FALCON microbatch: 256 candidates per user against L = randint(7372, 8192) cached rows, forward only, 24 layers back to back"),
    "L2048": (2048, 4, 128, 512, "long-sequence training point of the reference's own sweep (ops/benchmarks/hstu_attention_bench.py:139, seq_len 2^8..2^12): "
                                  "L = randint(1843, 2048), num_targets = randint(1, 21); the general multi-key-block backward"),
}


def make_lengths(workload, B, N, gen, device):
    if workload == "M-full":
        return torch.full((B,), N, dtype=torch.int64, device=device)
    if workload == "C2":
        return torch.randint(1, N + 1, (B,), generator=gen, device=device, dtype=torch.int64)
    if workload in ("C3", "C3-bias"):
        lengths = torch.randint(0, 30, (B,), generator=gen, device=device, dtype=torch.int64)
        full = torch.rand(B, generator=gen, device=device) < 0.05
        return torch.where(full, torch.full_like(lengths, N), lengths)
    lo = int(0.9 * N)  # generate_sparse_seq_len(sparsity=0.95): randint(int(0.9 N), N)
    return torch.randint(lo, N, (B,), generator=gen, device=device, dtype=torch.int64)


def _event_ms(pairs):
    return statistics.mean(a.elapsed_time(b) for a, b in pairs)


def attention_section(args, rank, world, device, telem=None):
    """the timed region of the headline number: K steps of attention fwd (+ bwd), bracketed by barrier + synchronize"""
    from generative_recommenders_amd import data_parallel as dp
    from generative_recommenders_amd.ops import _launch

    wl = args.workload
    N, H, d = args.max_seq_len, args.heads, args.head_dim
    B = args.users_per_gpu
    gen = torch.Generator(device=device).manual_seed(1001 + rank)
    lengths = make_lengths(wl if wl not in ("M-targets", "C4", "C5", "L2048") else "M-jag", B, N, gen, device)
    off = dp.local_offsets(lengths)
    L = int(off[-1].item())
    dtype = torch.bfloat16
    es = 2
    alpha = d**-0.5
    fwd_only = wl == "C5"
    kernels = {}

    if wl == "C5":
        delta, layers = 256, 24
        k = torch.empty(L, H, d, device=device, dtype=dtype).uniform_(-0.01, 0.01, generator=gen)
        v = torch.empty(L, H, d, device=device, dtype=dtype).uniform_(-0.01, 0.01, generator=gen)
        dq_ = torch.empty(B * delta, H, d, device=device, dtype=dtype).uniform_(-0.01, 0.01, generator=gen)
        nt = torch.full((B,), delta, dtype=torch.int64, device=device)

        def fwd():
            out = None
            for _ in range(layers):
                out = _launch.attn_fwd(dq_, k, v, off, nt, N, alpha, 1.0 / N, delta_q=delta)
            return out

        bwd = None
        fwd_bytes = layers * (L * H * 2 * d + 2 * B * delta * H * d) * es
        bwd_bytes = 0
        flops = layers * 4.0 * H * d * delta * float(L)            # every candidate row sees its user's whole history
        kernels["fwd"] = _launch.attn_fwd_kernel_name(dtype, d, d, N, heads=H)
        finite = lambda o: bool(torch.isfinite(o.float()).all())
    elif wl in ("C2", "C3-bias"):
        from generative_recommenders_amd.research.modeling.sequential import hstu as R

        torch.manual_seed(5)
        bias = R.RelativeBucketedTimeAndPositionBasedBias(N, 128).to(device)
        ts = torch.sort(torch.randint(0, 10**8, (B, N), generator=gen, device=device), dim=1).values
        q, k, v = (torch.empty(L, H * d, device=device, dtype=dtype).normal_(0, 0.3, generator=gen).requires_grad_() for _ in range(3))
        g = torch.randn(L, H * d, device=device, dtype=dtype, generator=gen)
        state = {}

        def fwd():
            state["out"] = R.hstu_rel_bias_attention(H, d, d, q, k, v, off, ts, N, bias)
            return state["out"]

        def bwd():
            return torch.autograd.grad(state["out"], (q, k, v, bias._pos_w, bias._ts_w), g)

        fwd_bytes = L * H * 4 * d * es + B * N * 8
        bwd_bytes = L * H * 7 * d * es + B * N * 8
        flops = float(sum(7.0 * H * d * float(x) * float(x) for x in lengths.tolist()))
        kernels["fwd"] = _launch.attn_fwd_kernel_name(dtype, d, d, N, heads=H, with_bias=True)
        kernels["bwd"] = _launch.attn_bwd_kernel_name(dtype, d, d, N, heads=H, with_bias=True)
        finite = lambda o: bool(torch.isfinite(o.float()).all())
    else:
        fused = torch.empty(L, H, 3 * d, device=device, dtype=dtype).uniform_(-0.01, 0.01, generator=gen)
        q, k, v = torch.split(fused, [d, d, d], dim=-1)
        if os.environ.get("HSTU_BENCH_LAYOUT") == "separate":   # experiment: contiguous (L, H, d) tensors
            q, k, v = (t.contiguous() for t in (q, k, v))
        elif os.environ.get("HSTU_BENCH_LAYOUT") == "headmajor":   # experiment: each head's rows contiguous (H, L, d)
            q, k, v = (t.permute(1, 0, 2).contiguous().permute(1, 0, 2) for t in (q, k, v))
        dout = torch.randn(L, H, d, device=device, dtype=dtype, generator=gen)
        nt = None
        if wl in ("M-targets", "L2048"):
            nt = torch.minimum(torch.randint(1, 21, (B,), generator=gen, device=device), lengths)
        dfused = torch.empty_like(fused)
        dq, dk, dv = torch.split(dfused, [d, d, d], dim=-1)

        # sort_by_length (reference default for the layer): workgroups take the users heaviest first
        order = _launch.length_order(off) if args.sort_by_length else None

        def fwd():
            return _launch.attn_fwd(q, k, v, off, nt, N, alpha, 1.0 / N, user_order=order)

        def bwd():
            return _launch.attn_bwd(dout, q, k, v, off, nt, N, alpha, 1.0 / N, dq=dq, dk=dk, dv=dv, user_order=order)

        fwd_bytes = L * H * (2 * d + 2 * d) * es
        bwd_bytes = L * H * (4 * d + 3 * d) * es
        # fwd+bwd, causal-halved (hstu_attention_bench.py:35-59): 4 f1 + 3 f2 per user
        flops = float(sum(7.0 * H * d * float(x) * float(x) for x in lengths.tolist())) if B <= 100000 else 0.0
        tg = dict(max_attn_len=0)
        kernels["fwd"] = _launch.attn_fwd_kernel_name(dtype, d, d, N, heads=H, alpha=alpha, **tg)
        kernels["bwd"] = _launch.attn_bwd_kernel_name(dtype, d, d, N, heads=H, alpha=alpha, **tg)
        finite = lambda o: bool(torch.isfinite(o.float()).all() and torch.isfinite(dfused.float()).all())

    # Pre-warm (untimed, in front of the W warm-up steps): the GPU sits idle while the process imports, builds inputs or -- between
    # the sections -- runs a CPU checker, and drops to its lowest power state; the first tens of milliseconds of load then include
    # the clock / power ramp, seen as ONE step of 40-50 ms among steps of 1-3 ms (profiles/r05_bench_b_default_forward_outlier.json,
    # M-jag in r05_bench_c_default.json).  Keep the kernels of this section running for ~0.3 s before anything is counted.
    prewarm_steps = 0
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < getattr(args, "prewarm_s", 0.3) and prewarm_steps < 400:
        fwd()
        if bwd is not None:
            bwd()
        prewarm_steps += 1
        if prewarm_steps % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    if telem is not None:
        telem.start()
    for _ in range(args.warmup):
        fwd()
        if bwd is not None:
            bwd()
    if telem is not None:
        torch.cuda.synchronize()
        telem.stop()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for i in range(args.steps):
        ev[i][0].record()
        out = fwd()
        ev[i][1].record()
        if bwd is not None:
            bwd()
        ev[i][2].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = dp.max_over_ranks(t1 - t0, device)
    fwd_ms = _event_ms([(e[0], e[1]) for e in ev])
    bwd_ms = _event_ms([(e[1], e[2]) for e in ev]) if bwd is not None else 0.0
    # spread of the per-step HIP-event durations (diagnostic only: `value` and the roofline use the K-step totals / means).
    # The forward of the first process on a fresh box has twice been seen at 2.2 ms instead of 1.37 for a whole run
    # (DESIGN 5): these fields tell a transient from a slow box.
    spread = lambda xs: dict(min=round(min(xs), 4), median=round(statistics.median(xs), 4), max=round(max(xs), 4))
    step_spread = {"fwd_ms": spread([e[0].elapsed_time(e[1]) for e in ev])}
    if bwd is not None:
        step_spread["bwd_ms"] = spread([e[1].elapsed_time(e[2]) for e in ev])
    assert finite(out), "non-finite values in the benchmark outputs"
    if telem is not None:      # untimed replay of the same steps under the sampler (see Telemetry)
        telem.start()
        for _ in range(min(args.steps, 25)):
            fwd()
            if bwd is not None:
                bwd()
        torch.cuda.synchronize()
        telem.stop()
    parity = None
    if getattr(args, "parity_users", 0) and wl in ("M-full", "M-jag", "M-targets", "C4") and rank == 0:
        try:
            parity = parity_sample(q, k, v, dout, out, dq, dk, dv, off, nt, N, alpha, n_users=args.parity_users)
        except Exception as e:  # pragma: no cover -- the headline number must survive a failure of the checker
            parity = {"error": repr(e)[:300]}
    return dict(
        parity=parity,
        elapsed=elapsed, users=B, rows=L, total_rows=dp.sum_over_ranks(float(L), device), fwd_ms=fwd_ms, bwd_ms=bwd_ms,
        fwd_gbps=fwd_bytes / fwd_ms / 1e6, bwd_gbps=(bwd_bytes / bwd_ms / 1e6) if bwd_ms else 0.0,
        both_gbps=(fwd_bytes + bwd_bytes) / (fwd_ms + bwd_ms) / 1e6, bwd_bytes=bwd_bytes, fwd_bytes=fwd_bytes, prewarm_steps=prewarm_steps,
        tflops=flops / ((fwd_ms + bwd_ms) * 1e-3) / 1e12, flops=flops, kernels=kernels, fwd_only=fwd_only,
        device_ms_per_step=fwd_ms + bwd_ms, step_spread=step_spread,
    )


def rooflines(att, workload):
    """roofline objects of one attention_section result.  Bound per workload: the jagged training shapes move ~64 FLOP per
    byte, far under the ridge (~310): HBM.  C5 (delta-q microbatches against 8 K cached rows) re-reads K/V from L2 for every
    query block and is bound by the matrix / vector pipes: its fraction is TFLOP/s of the causal FLOP model over the dense
    bf16 MFMA peak."""
    dom = "fwd" if att["fwd_only"] else "bwd"
    if workload == "C5":
        tf = att["tflops"]
        main = {"bound": "mfma", "kernel": att["kernels"][dom], "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / MFMA_PEAK_TFLOPS, "traffic": None, "algorithmic_bytes_per_launch": att[dom + "_bytes"],
                "avg_launch_ms": att[dom + "_ms"],
                "note": "VALU-limited in practice (two quarter-rate transcendentals per score); hbm-equivalent GB/s: %.0f" % att[dom + "_gbps"]}
    else:
        main = {"bound": "hbm", "kernel": att["kernels"][dom], "achieved": att[dom + "_gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": att[dom + "_gbps"] / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": att[dom + "_bytes"],
                "avg_launch_ms": att[dom + "_ms"]}
    fwd = {"bound": main["bound"], "kernel": att["kernels"]["fwd"], "achieved": att["fwd_gbps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": att["fwd_gbps"] / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": att["fwd_bytes"], "avg_launch_ms": att["fwd_ms"]}
    if workload == "C5":
        fwd = dict(main)
    both = {"achieved": att["both_gbps"], "unit": "GB/s", "frac": att["both_gbps"] / HBM_PEAK_GBPS, "tflops_causal_model": att["tflops"]}
    if workload == "L2048":
        # long sequences: ~900 FLOP per byte, far over the ridge: bound = the matrix pipes.  Causal FLOP model, 2 d L^2 forward
        # (two GEMMs over the triangle) + 5 d L^2 backward per user and head; the backward EXECUTES 7 d L^2 (the two-kernel schedule
        # recomputes S and dP in its dQ kernel: csrc/hstu_attn_bwd_long.cuh) -- the fraction prices the model's FLOPs, not those
        tf_f = att["flops"] * 2.0 / 7.0 / (att["fwd_ms"] * 1e-3) / 1e12
        tf_b = att["flops"] * 5.0 / 7.0 / (att["bwd_ms"] * 1e-3) / 1e12
        main = {"bound": "mfma", "kernel": att["kernels"]["bwd"], "achieved": tf_b, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf_b / MFMA_PEAK_TFLOPS,
                "traffic": None, "algorithmic_bytes_per_launch": att["bwd_bytes"], "avg_launch_ms": att["bwd_ms"]}
        fwd = {"bound": "mfma", "kernel": att["kernels"]["fwd"], "achieved": tf_f, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf_f / MFMA_PEAK_TFLOPS,
               "algorithmic_bytes_per_launch": att["fwd_bytes"], "avg_launch_ms": att["fwd_ms"]}
        both = {"achieved": att["tflops"], "unit": "TFLOP/s", "frac": att["tflops"] / MFMA_PEAK_TFLOPS, "tflops_causal_model": att["tflops"]}
    return main, fwd, both


# the other shapes north_star names, run for a few steps after the headline so that the driver's record carries them
# (M-full-1024: the metric shape at the PER-GPU batch of the 8-GPU metric -- 8192 users over 8 ranks: launch ramp, first prologue
# and last tail weigh 8x more than at 8192 users per GPU)
# (M-targets: what DLRM-v3 runs -- modules/dlrm_hstu.py:207-214 target_aware=True, sort_by_length=True; C4 / C5 / L2048: sub-10-ms or
# long steps, their own step counts)
EXTRA_WORKLOADS = [("M-jag", {}), ("M-targets", {"sort_by_length": True}),
                   ("M-full-1024", {"workload": "M-full", "users": 1024, "steps": 96, "warmup": 20}), ("M-full-d64", {"workload": "M-full", "head_dim": 64}),
                   ("C2", {}), ("C3", {}), ("C3-bias", {}), ("C4", {"steps": 48, "warmup": 10}), ("C5", {"steps": 3, "warmup": 1}),
                   ("L2048", {"steps": 6, "warmup": 2})]


def extra_workloads(args, rank, world, device):
    out = {}
    for name, over in EXTRA_WORKLOADS:
        a = argparse.Namespace(**vars(args))
        a.workload = over.get("workload", name)
        n, h, d, users, _ = WORKLOADS[a.workload]
        a.max_seq_len, a.heads, a.head_dim, a.users_per_gpu = n, h, over.get("head_dim", d), over.get("users", users)
        a.steps, a.warmup = over.get("steps", args.extra_steps), over.get("warmup", over.get("steps", 24) // 8)   # (sub-millisecond steps: more of them, or the loop is over before the clocks have settled)
        a.sort_by_length = over.get("sort_by_length", a.workload == "C3")
        a.parity_users = 0          # (the headline batch carries the oracle check)
        try:
            att = attention_section(a, rank, world, device)
            main, fwd, both = rooflines(att, a.workload)
            out[name] = {"user_seqs_per_s": world * att["users"] * a.steps / att["elapsed"], "steps": a.steps,
                         "users_per_gpu": a.users_per_gpu, "max_seq_len": n, "heads": h, "head_dim": a.head_dim,
                         "fwd_ms": round(att["fwd_ms"], 4), "bwd_ms": round(att["bwd_ms"], 4), "step_spread": att["step_spread"], "bound": main["bound"],
                         "prewarm_steps": att["prewarm_steps"], "warmup": a.warmup, "sort_by_length": bool(a.sort_by_length),
                         "frac_fwd": round(fwd["frac"], 4), "frac_bwd": round(main["frac"], 4), "frac_fwd_bwd": round(both["frac"], 4),
                         "kernels": att["kernels"], "what": WORKLOADS[a.workload][4]}
            # HBM bytes of the committed PMC passes of the same workload / kernels (None: no pass on this instantiation)
            fname, ent = traffic_entry(a.workload, a.users_per_gpu, a.head_dim, h)
            tr = {}
            for side in ("fwd", "bwd"):
                e = (ent or {}).get(side)
                if e and e.get("kernel") == att["kernels"].get(side):
                    tr[side] = {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"],
                                "over_algorithmic": round(e["hbm_bytes_per_launch"] / att[side + "_bytes"], 4)}
            out[name]["traffic"] = dict(tr, file="profiles/" + fname) if tr else None
            if ent is not None and ent.get("stale"):
                out[name]["traffic_stale"] = ent["stale"]["why"]
        except Exception as e:  # the headline number must survive a failure here
            out[name] = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
    return out


def copy_bandwidth(device):
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


class Telemetry:
    """Clocks / power / temperature of this rank's GPU sampled by a background thread, so that a slow box is visible in the
    bench line itself.  Source: the amdgpu sysfs nodes (hwmon freq1 = sclk, freq2 = mclk, power1, temp1..3; no subprocess);
    when they are not readable, one ``rocm-smi`` call before and after.  The sampler NEVER runs inside a timed region: it
    runs during the warm-up steps in front of it and during an untimed replay of the same steps behind it (the same kernels,
    the same load).  Sampling inside the timed loop was the first version: one run of three then showed a single forward
    step of 42 ms among fifty of 1.3 (gpurun r05b: the hwmon reads go through the SMU) -- 17 % of the headline from one
    hiccup."""

    def __init__(self, device, period=0.02):
        import glob
        import threading

        self.period, self.samples, self._stop, self._thread = period, [], threading.Event(), None
        self.nodes, self.source = {}, None
        bus = None
        try:
            pr = torch.cuda.get_device_properties(device)
            bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        cards = []
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            try:
                if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                    continue
                slot = os.path.basename(os.path.realpath(dev))
                cards.append((slot, dev))
            except Exception:
                continue
        pick = [c for c in cards if bus and c[0].startswith(bus)] or cards[:1]
        if pick:
            dev = pick[0][1]
            hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
            if hw:
                for key, names in (("sclk_mhz", ("freq1_input",)), ("mclk_mhz", ("freq2_input",)),
                                   ("power_w", ("power1_average", "power1_input")),
                                   ("temp_edge_c", ("temp1_input",)), ("temp_junction_c", ("temp2_input",)), ("temp_mem_c", ("temp3_input",))):
                    for n in names:
                        f = os.path.join(hw[0], n)
                        if os.access(f, os.R_OK):
                            self.nodes[key] = f
                            break
            self.source = f"sysfs {pick[0][0]}" + ("" if bus and pick[0][0].startswith(bus) else " (first amdgpu card: PCI id of the torch device not matched)")
        self._smi = None if self.nodes else self._rocm_smi()

    @staticmethod
    def _rocm_smi():
        import subprocess

        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
            card = next(iter(json.loads(out[out.index("{"):]).values()))
            return {k: v for k, v in card.items() if any(w in k.lower() for w in ("sclk", "mclk", "power", "temperature"))}
        except Exception as e:
            return {"error": repr(e)[:200]}

    def _read(self):
        row = {}
        for key, f in self.nodes.items():
            try:
                v = float(open(f).read().strip())
                row[key] = v / 1e6 if key.endswith("_mhz") or key == "power_w" else v / 1e3
            except Exception:
                pass
        return row

    def start(self):
        """sample until stop(); may be called several times (the samples accumulate)"""
        import threading

        if self.nodes and self._thread is None:
            self._stop.clear()

            def loop():
                while not self._stop.is_set():
                    self.samples.append(self._read())
                    self._stop.wait(self.period)
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None

    def __enter__(self):
        self.start()
        return self

    def __exit__(self, *exc):
        self.stop()
        return False

    def summary(self):
        if not self.nodes:
            return {"source": "rocm-smi (sysfs hwmon not readable)", "before": self._smi, "after": self._rocm_smi()}
        res = {"source": self.source, "samples": len(self.samples), "period_s": self.period}
        for key in self.nodes:
            xs = [r[key] for r in self.samples if key in r]
            if xs:
                res[key] = {"min": round(min(xs), 1), "mean": round(sum(xs) / len(xs), 1), "max": round(max(xs), 1)}
        return res


def calibration(device):
    """what THIS box sustains for the two resources the kernels are priced against, from streams that do nothing else
    (csrc/aux_ops.hip): an MFMA chain without memory traffic (two waves per SIMD, 32x32x16 bf16) and a non-temporal
    read of 2 GiB; plus the plain device copy torch issues.  The product numbers of two boxes compare through these."""
    from generative_recommenders_amd.ops import _launch

    res = {}
    launch, flop = _launch.calib_mfma_stream(device, iters=8192)
    for _ in range(2):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        launch()
    e1.record()
    torch.cuda.synchronize()
    tf = flop * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    # 1024 FLOP per clock and SIMD (32x32x16 bf16: 32768 FLOP in 8 passes of 4 clocks): the clock an MFMA-saturated chip holds
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    res["mfma_stream_tflops"] = round(tf, 1)
    res["mfma_stream_frac_of_2500"] = round(tf / MFMA_PEAK_TFLOPS, 3)
    res["mfma_stream_effective_mhz"] = round(tf * 1e12 / (n_cu * 4 * 1024) / 1e6, 0)
    buf = torch.empty(2 << 30, dtype=torch.uint8, device=device)
    rd = _launch.calib_read_stream(buf)
    for _ in range(2):
        rd()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        rd()
    e1.record()
    torch.cuda.synchronize()
    res["read_stream_GBps"] = round(buf.numel() * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 0)
    del buf
    res["copy_GBps"] = round(copy_bandwidth(device), 0)
    return res


def parity_sample(q, k, v, dout, out, dq, dk, dv, off, nt, N, alpha, n_users=32, seed=77):
    """``n_users`` randomly chosen users of the batch the timed loop ran on: out, dq, dk, dv as the kernels left them
    against oracle/hstu_oracle.py (fp64) on the same rows -- relative Frobenius error per tensor.  Runs AFTER the timed
    region; the oracle is the checker only."""
    import numpy as np

    from oracle import hstu_oracle as O

    B = off.numel() - 1
    gen = torch.Generator().manual_seed(seed)
    users = torch.randperm(B, generator=gen)[:min(n_users, B)].sort().values
    off_c = off.cpu()
    rows = torch.cat([torch.arange(int(off_c[u]), int(off_c[u + 1])) for u in users.tolist()]).to(q.device)
    lens = torch.tensor([int(off_c[u + 1] - off_c[u]) for u in users.tolist()])
    so = torch.zeros(len(lens) + 1, dtype=torch.int64)
    so[1:] = torch.cumsum(lens, 0)
    f = lambda t: t.index_select(0, rows).double().cpu().numpy()
    nts = None if nt is None else nt.index_select(0, users.to(nt.device)).cpu().numpy()
    qs, ks, vs, dos = f(q), f(k), f(v), f(dout)
    ref_o = O.hstu_mha_fwd(N, alpha, qs, ks, vs, so.numpy(), num_targets=nts)
    ref_q, ref_k, ref_v = O.hstu_mha_bwd(N, alpha, dos, qs, ks, vs, so.numpy(), num_targets=nts)
    res = {}
    for name, got, want in (("out", out, ref_o), ("dq", dq, ref_q), ("dk", dk, ref_k), ("dv", dv, ref_v)):
        g = f(got)
        res[name] = float(np.linalg.norm(g - want) / max(np.linalg.norm(want), 1e-300))
    worst = max(res.values())
    return {"users_checked": int(len(lens)), "rows_checked": int(so[-1]), "rel_fro": {k_: float("%.3e" % v_) for k_, v_ in res.items()},
            "max_rel_fro": float("%.3e" % worst), "gate_bf16": 3.8e-3, "ok": bool(worst < 3.8e-3),
            "what": "users drawn at random (seed 77) from the timed batch, outputs of the LAST timed step against oracle/hstu_oracle.py in fp64 "
                    "(gate = tests/test_attention_gpu.py's bf16 gate; 1.66e-3 of it is the rounding of an exact result to bf16)"}


def rccl_section(world, device, nbytes=22 << 20):
    """communicator size + bus bandwidth of the all-reduce the layer step performs (22 MB of fp32 gradients); with ONE rank
    (a one-rank RCCL communicator on a one-GPU box) the time of the library's launch path, bus bandwidth 0 by definition"""
    if not dist.is_initialized():
        return None
    t = torch.ones(nbytes // 4, dtype=torch.float32, device=device)
    for _ in range(3):
        dist.all_reduce(t)
    if device.type == "cuda":
        torch.cuda.synchronize()
    dist.barrier()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(t)
    if device.type == "cuda":
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    from generative_recommenders_amd import data_parallel as dp

    dt = dp.max_over_ranks(dt, device)
    return dict(backend=dist.get_backend(), ranks=dist.get_world_size(), allreduce_bytes=nbytes, allreduce_ms=dt * 1e3,
                busbw_GBps=2.0 * (world - 1) / world * nbytes / dt / 1e9)


def layer_section(args, rank, world, device, telem=None):
    """3 x STULayer (D=512, H=4, dqk=dv=128, group norm, target-aware) fwd+bwd + gradient
    all-reduce, bf16 activations (DLRM-v3 HSTU config, dlrm_v3/configs.py:30-41)."""
    from generative_recommenders_amd import data_parallel as dp
    from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig, STUStack

    N, H, d = 200, 4, 128
    D = H * d
    B = args.layer_users_per_gpu
    gen = torch.Generator(device=device).manual_seed(2002 + rank)
    lengths = make_lengths("M-jag", B, N, gen, device)
    off = dp.local_offsets(lengths)
    L = int(off[-1].item())
    x = torch.randn(L, D, device=device, dtype=torch.bfloat16, generator=gen).requires_grad_()
    gy = torch.randn(L, D, device=device, dtype=torch.bfloat16, generator=gen)
    nt = torch.minimum(torch.randint(1, 21, (B,), generator=gen, device=device), lengths)

    def timed(recompute, dropout, fuse=True, telem=None, replay=0, force_collectives=False):
        """recompute=True: the reference's STULayerConfig defaults (normed x, uvqk and y recomputed in the backward --
        a memory saving sized for 80 GB parts); False: everything kept (3 layers x 1024 users: 2.4 GB of 288).
        dropout: output_dropout_ratio of the layers (DLRM-v3 trains with hstu_linear_dropout_rate = 0.1,
        dlrm_v3/configs.py:39), fused into the norm kernel, mask regenerated in backward."""
        torch.manual_seed(7)
        stack = STUStack([STULayer(STULayerConfig(embedding_dim=D, num_heads=H, hidden_dim=d, attention_dim=d,
                                                  output_dropout_ratio=dropout, use_group_norm=True, recompute_normed_x=recompute,
                                                  recompute_uvqk=recompute, recompute_y=recompute)) for _ in range(3)]).to(device)
        stack.train()
        for layer in stack._stu_layers:
            layer.fuse_layer = fuse        # True: the layer as one autograd node (default); False: the reference's two nodes
        # one bucket per layer, its all-reduce launched from inside backward when the layer's last gradient is in
        reducer = dp.GradientAllReducer(None, buckets=[layer.parameters() for layer in stack._stu_layers], overlap=True,
                                        single_rank_collectives=force_collectives)

        def step():
            for p in stack.parameters():
                p.grad = None
            x.grad = None
            y = stack(x=x, x_lengths=lengths, x_offsets=off, max_seq_len=N, num_targets=nt)
            y.backward(gy)
            reducer.reduce()

        # pre-warm, untimed (see attention_section): a FIXED number of the same steps -- they contain the gradient all-reduce, so
        # every rank must run the same count (20 steps ~ 0.27 s)
        for _ in range(20 if getattr(args, "prewarm_s", 0.3) > 0 else 0):
            step()
        torch.cuda.synchronize()
        if telem is not None:
            telem.start()
        for _ in range(3):
            step()
        if telem is not None:
            torch.cuda.synchronize()
            telem.stop()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.layer_steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = dp.max_over_ranks(time.perf_counter() - t0, device)
        if replay:      # untimed replay under rank 0's sampler (see Telemetry).  EVERY rank runs it: the steps contain collectives
            if telem is not None:
                telem.start()
            for _ in range(replay):
                step()
            torch.cuda.synchronize()
            if telem is not None:
                telem.stop()
        return elapsed, stack

    p_drop = args.layer_dropout
    if getattr(args, "layer_tunableop", False):
        import torch.cuda.tunable as tunable

        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(200)
    elapsed_keep, _ = timed(False, p_drop)
    elapsed_nodrop, _ = timed(True, 0.0)
    elapsed_two, _ = timed(True, p_drop, fuse=False)
    elapsed, stack = timed(True, p_drop, telem=telem, replay=min(args.layer_steps, 5))
    nparams = sum(p.numel() for p in stack.parameters())
    one_rank = None
    if world == 1 and dist.is_initialized():
        # the same step with the buckets REALLY all-reduced on the one-rank RCCL communicator (hooks -> staging copy -> asynchronous
        # ncclAllReduce on the communicator's stream -> p.grad): the collective path's cost on this box, not part of ms_per_step
        try:
            e1, _ = timed(True, p_drop, force_collectives=True)
            one_rank = dict(ms_per_step=e1 / args.layer_steps * 1e3, backend=dist.get_backend(),
                            what="layer step with every bucket all-reduced on a ONE-rank communicator (identity sums; tests/test_rccl_gpu.py "
                                 "checks them bit for bit)")
        except Exception as e:  # pragma: no cover
            one_rank = {"error": repr(e)[:300]}
    gemm_flops = 3 * 3 * L * (2 * D * 4 * D + 2 * 3 * D * D)  # 3 layers x (fwd + 2x bwd) x (uvqk + output)
    return dict(users_per_gpu=B, steps=args.layer_steps, ms_per_step=elapsed / args.layer_steps * 1e3,
                user_seqs_per_s=world * B * args.layer_steps / elapsed, params=nparams,
                gemm_selection="TunableOp" if getattr(args, "layer_tunableop", False) else "hipBLASLt heuristic (default)",
                config=f"3 STU layers D=512, 4 heads of 128, group norm, targets; training mode, output_dropout_ratio={p_drop} "
                       f"(fused, mask regenerated in backward), recompute normed_x / uvqk / y in backward; gradient all-reduce: "
                       f"one bucket per layer launched from backward hooks; each layer ONE autograd node (SiLU and the residual's "
                       f"gradient inside the neighbouring row kernels)",
                two_node_layers=dict(ms_per_step=elapsed_two / args.layer_steps * 1e3,
                                     user_seqs_per_s=world * B * args.layer_steps / elapsed_two),
                dropout_off=dict(ms_per_step=elapsed_nodrop / args.layer_steps * 1e3,
                                 user_seqs_per_s=world * B * args.layer_steps / elapsed_nodrop),
                allreduce_bytes=nparams * 4, one_rank_collectives=one_rank,
                no_recompute=dict(ms_per_step=elapsed_keep / args.layer_steps * 1e3,
                                  user_seqs_per_s=world * B * args.layer_steps / elapsed_keep),
                gemm_mfma_frac_if_all_time_were_gemm=gemm_flops * args.layer_steps / elapsed / 1e12 / MFMA_PEAK_TFLOPS,
                projections=projection_section(L, D, device))


def projection_section(rows, D, device):
    """MFMA utilisation of the projections of one STU layer at the layer section's shape, HIP events around 10 calls each:
    TFLOP/s and the fraction of the dense bf16 peak.  ``uvqk_fwd_fused`` / ``uvqk_fwd_fused_with_normed`` = what the product
    runs for the forward UVQK projection (and its recompute in backward): the hand-written LayerNorm + GEMM kernel
    (csrc/hstu_ln_linear.cuh; the layer norm is INSIDE the timed call; also us per call and GB/s of its algorithmic bytes --
    x in, y out -- against 8 TB/s).  ``uvqk_fwd`` = the hipBLASLt ``linear`` the product falls back to for shapes the fused
    kernel does not take, kept as the comparator (no layer norm in it).  The other five are the calls ops/hstu_compute.py
    makes: addmm with the residual, mm for the data gradients, ops/mm.py's slab-split batched GEMM for the weight gradients;
    ``bias_grad`` = hstu_column_sum (HBM-bound: GB/s)."""
    from generative_recommenders_amd.ops import _launch
    from generative_recommenders_amd.ops.hstu_compute import _uvqk_dgrad, _uvqk_gemm, _uvqk_prepare
    from generative_recommenders_amd.ops.mm import weight_grad_mm

    dt = torch.bfloat16
    x = torch.randn(rows, D, device=device, dtype=dt)
    w_uvqk = torch.randn(D, 4 * D, device=device, dtype=dt)
    b_uvqk = torch.randn(4 * D, device=device, dtype=dt)
    g_uvqk = torch.randn(rows, 4 * D, device=device, dtype=dt)
    y3 = torch.randn(rows, 3 * D, device=device, dtype=dt)
    w_out = torch.randn(3 * D, D, device=device, dtype=dt)
    g_out = torch.randn(rows, D, device=device, dtype=dt)
    w_mul, kmajor = _uvqk_prepare(w_uvqk, dt)      # the K-major copy the product caches per parameter version
    ln_w, ln_b = torch.ones(D, device=device, dtype=dt), torch.zeros(D, device=device, dtype=dt)
    fused_ok = kmajor and _launch.ln_linear_supported(x, 4 * D)
    cases = {}
    if fused_ok:
        cases["uvqk_fwd_fused"] = (lambda: _launch.ln_linear_fwd(x, ln_w, ln_b, 1e-6, w_mul, b_uvqk, want_normed=False), 2.0 * rows * D * 4 * D)
        cases["uvqk_fwd_fused_with_normed"] = (lambda: _launch.ln_linear_fwd(x, ln_w, ln_b, 1e-6, w_mul, b_uvqk, want_normed=True), 2.0 * rows * D * 4 * D)
    cases.update({
        "uvqk_fwd": (lambda: _uvqk_gemm(x, w_mul, kmajor, b_uvqk), 2.0 * rows * D * 4 * D),
        "uvqk_dgrad": (lambda: _uvqk_dgrad(g_uvqk, w_mul, kmajor), 2.0 * rows * D * 4 * D),
        "uvqk_wgrad": (lambda: weight_grad_mm(x, g_uvqk), 2.0 * rows * D * 4 * D),
        "out_fwd": (lambda: torch.addmm(x, y3, w_out), 2.0 * rows * 3 * D * D),      # + the residual: copy of x + in-place GEMM
        "out_dgrad": (lambda: torch.mm(g_out, w_out.t()), 2.0 * rows * 3 * D * D),
        "out_wgrad": (lambda: weight_grad_mm(y3, g_out), 2.0 * rows * 3 * D * D),
    })
    lt_ok = _launch.addmm_residual_supported(x, y3, w_out)
    if lt_ok:
        cases["out_fwd_one_launch"] = (lambda: _launch.addmm_residual(x, y3, w_out), 2.0 * rows * 3 * D * D)
    k512_ok = _launch.linear_k512_supported(g_out, 3 * D)
    if k512_ok:
        cases["out_dgrad_k512"] = (lambda: _launch.linear_k512(g_out, w_out), 2.0 * rows * 3 * D * D)
    hbm_bytes = {"uvqk_fwd_fused": rows * (D + 4 * D) * 2.0, "uvqk_fwd_fused_with_normed": rows * (2 * D + 4 * D) * 2.0,
                 "bias_grad": rows * 4 * D * 2.0}
    if _launch.column_sum_supported(g_uvqk):
        cases["bias_grad"] = (lambda: _launch.column_sum(g_uvqk), 0.0)
    res = {}
    for name, (fn, flops) in cases.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        tf = flops / (us * 1e-6) / 1e12
        res[name] = {"tflops": round(tf, 1), "mfma_frac": round(tf / MFMA_PEAK_TFLOPS, 3), "us": round(us, 1)}
        if name in hbm_bytes:
            gbps = hbm_bytes[name] / (us * 1e-6) / 1e9
            res[name].update(algorithmic_GBps=round(gbps, 0), hbm_frac=round(gbps / HBM_PEAK_GBPS, 3))
        if name.startswith("uvqk_fwd_fused"):
            res[name]["kernel"] = "hstu_ln_linear_fwd_kernel (LayerNorm inside; what the product runs)"
        if name == "out_fwd_one_launch":
            res[name]["kernel"] = "hipBLASLt matmul with separate C and D (hstu_addmm_residual; what the product runs)"
        if name == "out_fwd":
            res[name]["kernel"] = "torch.addmm: copy of x + hipBLASLt in place (comparator: the product runs out_fwd_one_launch)" if lt_ok else "torch.addmm"
        if name == "out_dgrad_k512":
            res[name]["kernel"] = "hstu_ln_linear_fwd_kernel without the LayerNorm (hstu_linear_k512; what the product runs)"
        if name == "out_dgrad":
            res[name]["kernel"] = "hipBLASLt mm (comparator: the product runs out_dgrad_k512 at this shape)" if k512_ok else "hipBLASLt mm"
        if name == "uvqk_fwd":
            res[name]["kernel"] = "hipBLASLt linear (comparator: the product runs uvqk_fwd_fused at this shape)" if fused_ok else "hipBLASLt linear"
    return res


def cpu_baseline(args):
    """The reference's padded-dense PyTorch algorithm (ported: oracle/dense_torch.py) fwd+bwd on
    the host cores, fp32, on a bounded sample of the same workload."""
    from oracle.dense_torch import dense_hstu_mha

    # SURVEY 8(d) names the reference's own hstu_mha(kernel=HammerKernel.PYTORCH).  It is Python under /root/reference, which exists
    # in the build container only (and may not travel to the GPU box): where it is importable it IS what gets timed (kind
    # "reference"); everywhere else the port below, with the measured port / reference ratio of the same sample next to it
    # (profiles/r06_cpu_reference_vs_port.json, tools/cpu_reference_vs_port.py: identical results, 0.93-1.25x the speed).
    ref_mha = None
    if os.path.isdir("/root/reference/generative_recommenders") and os.environ.get("HSTU_BENCH_CPU_REFERENCE", "1") != "0":
        try:
            sys.path.insert(0, "/root/reference")
            sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
            import _fbgemm_shim  # noqa: F401
            from generative_recommenders.common import HammerKernel
            from generative_recommenders.ops.hstu_attention import hstu_mha as _ref

            ref_mha = lambda n, a, q_, k_, v_, o_: _ref(max_seq_len=n, alpha=a, q=q_, k=k_, v=v_, seq_offsets=o_, causal=True,
                                                       dropout_pr=0.0, training=True, kernel=HammerKernel.PYTORCH)
        except Exception:
            ref_mha = None
    # more threads than this only add synchronisation cost on a (256, 4, 200, 200) problem
    cores = min(len(os.sched_getaffinity(0)), args.cpu_threads)
    torch.set_num_threads(cores)
    N, H, d = args.max_seq_len, args.heads, args.head_dim
    B = args.cpu_users
    gen = torch.Generator().manual_seed(1001)
    lengths = make_lengths(args.workload if args.workload in ("M-full", "C2", "C3", "C3-bias") else "M-jag", B, N, gen, "cpu")
    off = torch.zeros(B + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lengths, 0)
    L = int(off[-1])
    q = torch.empty(L, H, d).uniform_(-0.01, 0.01, generator=gen).requires_grad_()
    k = torch.empty(L, H, d).uniform_(-0.01, 0.01, generator=gen).requires_grad_()
    v = torch.empty(L, H, d).uniform_(-0.01, 0.01, generator=gen).requires_grad_()
    do = torch.randn(L, H, d, generator=gen)
    times = []
    # one warm-up pass, then timed passes over the same chunk until ~10 s of CPU work are on the clock
    # (bounded: the default bench run must stay within minutes whatever the host is)
    while len(times) < 2 or (sum(times[1:]) < 10.0 and len(times) < 41):
        t0 = time.perf_counter()
        out = ref_mha(N, d**-0.5, q, k, v, off) if ref_mha is not None else dense_hstu_mha(N, d**-0.5, q, k, v, off)
        out.backward(do)
        times.append(time.perf_counter() - t0)
        q.grad = k.grad = v.grad = None
    med = statistics.median(times[1:])
    res = dict(value=B / med, unit="user-seqs/s", cores=cores, kind="reference" if ref_mha is not None else "port",
               sample=f"{B} users of the same length distribution, fp32, fwd+bwd, median of {len(times) - 1} passes "
                      f"after 1 warm-up ({med * 1e3:.0f} ms each, {sum(times[1:]):.1f} s of CPU work); "
                      + ("the reference's hstu_mha(kernel=HammerKernel.PYTORCH), imported from /root/reference" if ref_mha is not None else
                         "oracle/dense_torch.py = reference pt_hstu_attention.py algorithm (the reference itself is not on this box)"))
    try:
        rv = json.load(open(os.path.join(ROOT, "profiles", "r06_cpu_reference_vs_port.json")))
        w = rv["workloads"].get(args.workload) or rv["workloads"]["M-jag"]
        res["port_over_reference"] = {"ratio": round(w["port_over_reference"], 3), "threads": rv["threads"], "where": rv["host"],
                                      "max_rel_diff_of_results": w["max_rel_diff_of_results"],
                                      "file": "profiles/r06_cpu_reference_vs_port.json"}
    except Exception:
        pass
    try:
        res["layer"] = cpu_baseline_layer(args, cores)
    except Exception as e:  # pragma: no cover
        res["layer"] = {"error": repr(e)[:300]}
    return res


def cpu_baseline_layer(args, cores):
    """3 STU layers (D=512, 4 heads of 128, group norm, targets) fwd+bwd with the reference's PyTorch-path algorithm on
    the host cores (oracle/dense_torch.py::dense_stu_stack), fp32, 64 users of the layer section's length distribution"""
    from oracle.dense_torch import dense_stu_stack

    N, H, d = 200, 4, 128
    D = H * d
    B = 64
    gen = torch.Generator().manual_seed(2002)
    lengths = make_lengths("M-jag", B, N, gen, "cpu")
    off = torch.zeros(B + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lengths, 0)
    L = int(off[-1])
    layers = []
    for _ in range(3):
        prm = {"_input_norm_weight": torch.ones(D), "_input_norm_bias": torch.zeros(D),
               "_uvqk_weight": torch.randn(D, 4 * D, generator=gen) * 0.02, "_uvqk_beta": torch.zeros(4 * D),
               "_output_norm_weight": torch.ones(H), "_output_norm_bias": torch.zeros(H),
               "_output_weight": torch.randn(3 * D, D, generator=gen) * 0.02}
        for t in prm.values():
            t.requires_grad_()
        layers.append((prm, True))
    x = torch.randn(L, D, generator=gen).requires_grad_()
    gy = torch.randn(L, D, generator=gen)
    nt = torch.minimum(torch.randint(1, 21, (B,), generator=gen), lengths)
    times = []
    while len(times) < 2 or (sum(times[1:]) < 8.0 and len(times) < 21):
        t0 = time.perf_counter()
        y = dense_stu_stack(x, layers, num_heads=H, attn_dim=d, hidden_dim=d, max_seq_len=N, seq_offsets=off, num_targets=nt)
        y.backward(gy)
        times.append(time.perf_counter() - t0)
        x.grad = None
        for prm, _ in layers:
            for t in prm.values():
                t.grad = None
    med = statistics.median(times[1:])
    return dict(value=B / med, unit="user-seqs/s", cores=cores, kind="port",
                sample=f"{B} users, 3 STU layers D=512 fp32 fwd+bwd, median of {len(times) - 1} passes after 1 warm-up "
                       f"({med * 1e3:.0f} ms each); oracle/dense_torch.py::dense_stu_stack = reference modules/stu.py PyTorch path")


TRAFFIC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json")

# The HBM-traffic figures are counters of committed rocprofv3 passes, not of this run: they describe THE KERNELS THAT WERE PROFILED.
# Every entry therefore carries the SHA-256 of the attention kernels' sources it was taken on (``sources_sha256``, written by
# tools/make_pmc_traffic.py); an entry without one, or with another one, is STALE -- the kernels were edited after the passes --
# and is refused: ``traffic`` stays null and ``traffic_stale`` says why.
KERNEL_SOURCE_GLOBS = ("hstu_attn_*.cuh", "hstu_common.cuh", "attn_*.cuh", "attn_*.hip", "capi_internal.h")


def kernel_sources_sha256():
    import glob
    import hashlib

    h = hashlib.sha256()
    base = os.path.join(ROOT, "generative_recommenders_amd", "csrc")
    files = sorted({f for g in KERNEL_SOURCE_GLOBS for f in glob.glob(os.path.join(base, g))})
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def traffic_entry(workload, users, head_dim, heads):
    """(file, entry) of the committed PMC passes of this (workload, users, head dim, heads) at bf16 taken on the CURRENT kernel
    sources; a matching entry of other sources comes back as (file, {"stale": reason}); (None, None) when there is none"""
    cur = None
    stale = (None, None)
    for name in TRAFFIC_FILES:
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        for ent in tr.get("entries", [tr]):
            src = ent.get("source", {"workload": "M-full", "users_per_gpu": 8192, "head_dim": 128, "heads": 4, "dtype": "bf16"})
            if (src.get("workload") == workload and src.get("users_per_gpu") == users and src.get("head_dim") == head_dim
                    and src.get("heads") == heads and src.get("dtype") == "bf16"):
                cur = cur or kernel_sources_sha256()
                sha = ent.get("sources_sha256", tr.get("sources_sha256"))
                if sha == cur:
                    return name, ent
                if stale[0] is None:
                    stale = (name, {"stale": {"file": "profiles/" + name, "why": "the attention kernel sources changed after these PMC passes were taken"
                                              if sha else "entry carries no hash of the kernel sources it was taken on",
                                              "entry_sources_sha256": sha, "current_sources_sha256": cur}})
    return stale


def attach_traffic(res, args, att):
    """HBM bytes from committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, collected with tools/prof_traffic.sh, written
    by tools/make_pmc_traffic.py; counters cannot be read from inside the process) -- attached only when the run's workload,
    users, head dim and dtype are the ones the passes were taken on AND the kernel sources are the ones that were profiled"""
    res["roofline"]["traffic"] = None
    name, ent = traffic_entry(args.workload, args.users_per_gpu, args.head_dim, args.heads)
    if ent is None:
        return
    if ent.get("stale"):
        res["roofline"]["traffic_stale"] = ent["stale"]
        return
    src = ent.get("source", {})
    dom = ent.get("bwd", ent["fwd"])            # forward-only workloads: the forward is the dominant kernel
    if dom.get("kernel") not in (res["roofline"].get("kernel"), None):
        return                                  # the passes were taken on another instantiation: not this run's traffic
    per = dom["hbm_bytes_per_launch"]
    res["roofline"]["traffic"] = per
    res["roofline"]["traffic_over_algorithmic"] = per / res["roofline"]["algorithmic_bytes_per_launch"]
    res["roofline"]["traffic_source"] = dict(file="profiles/" + name, kernel=dom.get("kernel"), sources_sha256=ent.get("sources_sha256"), **src)
    if "roofline_fwd" in res and "bwd" in ent:
        res["roofline_fwd"]["traffic"] = ent["fwd"]["hbm_bytes_per_launch"]
        res["roofline_fwd"]["traffic_source"] = dict(file="profiles/" + name, kernel=ent["fwd"].get("kernel"), **src)


def selftest_dist(args, rank, world, out):
    """the N-rank scaffolding without the kernels (CPU, gloo): barrier-bracketed timing, MAX over ranks, the all-reduce
    probe and the one-line JSON contract -- what tests/test_data_parallel.py runs with --gpus 2"""
    from generative_recommenders_amd import data_parallel as dp

    device = torch.device("cpu")
    x = torch.randn(256, 256)
    for _ in range(args.warmup):
        x = torch.tanh(x @ x.t() / 256)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = torch.tanh(x @ x.t() / 256)
    if world > 1:
        dist.barrier()
    elapsed = dp.max_over_ranks(time.perf_counter() - t0, device)
    res = {"metric": "selftest (no kernels)", "value": world * args.steps / elapsed, "unit": "steps/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "selftest"},
           "selftest": True, "rccl": rccl_section(world, device, nbytes=1 << 20),
           "ranks_seen": int(dp.sum_over_ranks(1.0, device))}
    if rank == 0:
        out.emit(json.dumps(res))


class _StdoutToStderr:
    """The contract is ONE JSON line on stdout.  Libraries write there too -- RCCL prints its version banner to stdout when the box
    exports NCCL_DEBUG=VERSION (the GPU boxes of this build do), hipBLASLt and MIOpen have their own moods -- so for the whole run
    file descriptor 1 points at stderr, and the line goes to the saved descriptor at the very end."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    @staticmethod
    def _flush_c_stdio():
        # (C stdio is fully buffered into a pipe or a file: what RCCL printf'ed would otherwise leave at process exit -- AFTER descriptor
        # 1 is back in place, i.e. behind the JSON line on stdout)
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    def emit(self, text):
        sys.stdout.flush()
        self._flush_c_stdio()
        os.write(self.saved, (text + "\n").encode())

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._flush_c_stdio()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def run(args):
    with _StdoutToStderr() as out:
        _run(args, out)


def _run(args, out):
    from generative_recommenders_amd import data_parallel as dp

    if os.environ.get("HSTU_BENCH_WATCHDOG"):      # debugging aid: dump every thread's stack and exit if the run stalls
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["HSTU_BENCH_WATCHDOG"]), exit=True)

    if args.selftest_dist:
        rank, local_rank, world = dp.init_from_env(backend="gloo")
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        selftest_dist(args, rank, world, out)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    from generative_recommenders_amd import _lib

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HSTU ops are HIP kernels with no CPU fallback")
    # The CPU baselines (10 + 8 s of host work on rank 0) run BEFORE the process group exists: the other ranks wait in the
    # rendezvous of init_process_group (a store wait), not inside a collective under the RCCL watchdog.
    cpu_res = None
    if int(os.environ.get("RANK", "0")) == 0 and not args.no_cpu:
        try:
            cpu_res = cpu_baseline(args)
        except Exception as e:  # pragma: no cover
            cpu_res = {"error": repr(e)[:300]}
        torch.set_num_threads(max(1, min(8, len(os.sched_getaffinity(0)))))
    # (--gpus 1: a ONE-rank RCCL communicator -- the library, its version string and the reducer's stream discipline on the one GPU
    # there is: the layer section's gradient buckets then really go through ncclAllReduce; HSTU_BENCH_SINGLE_RANK_RCCL=0 turns it off)
    single = int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("HSTU_BENCH_SINGLE_RANK_RCCL", "1") != "0"
    if single:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
    try:
        rank, local_rank, world = dp.init_from_env(single_rank_group=single)
    except Exception as e:  # pragma: no cover -- the headline number must survive a communicator that does not come up
        if not single:
            raise
        print(f"bench.py: one-rank process group not available ({e!r}); continuing without", file=sys.stderr)
        rank, local_rank, world = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one device per rank (init_from_env refuses anything else under RCCL); the modulo only serves the gloo rehearsal of the
    # multi-rank flow on a one-GPU box (HSTU_DIST_BACKEND=gloo)
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    _lib.lib()

    telem = Telemetry(device) if rank == 0 else None
    att = attention_section(args, rank, world, device, telem)
    value = world * att["users"] * args.steps / att["elapsed"]
    N, H, d = args.max_seq_len, args.heads, args.head_dim
    what = "fwd" if att["fwd_only"] else "fwd+bwd"
    res = {
        "metric": f"user-seqs/sec ({what}) HSTU attention L={N} d={H * d}",
        "value": value,
        "unit": "user-seqs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": att["elapsed"] / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {args.users_per_gpu} users/GPU, L<= {N}, H={H}, dqk=dv={d}; {WORKLOADS[args.workload][4]}; "
                        f"attention {what} via the C ABI" + ("" if args.workload in ("C2", "C3-bias", "C5") else ", q/k/v strided views of one fused buffer"),
            "users_per_gpu": args.users_per_gpu, "rows_per_gpu": att["rows"], "sort_by_length": args.sort_by_length, "parallelism": f"dp{world} (no collective: attention has no parameters)",
        },
        "device_ms_per_step": att["device_ms_per_step"], "step_spread": att["step_spread"],
        "prewarm_steps": att["prewarm_steps"], "prewarm_s": getattr(args, "prewarm_s", 0.3),
    }
    res["roofline"], res["roofline_fwd"], res["roofline_fwd_bwd"] = rooflines(att, args.workload)
    res["parity_at_this_size"] = att["parity"] if att.get("parity") else (
        "by invariance only: batch-composition bit-exactness, delta == tail of full, linearity in V "
        "(tests/test_attention_gpu.py); the reference-minted vectors at this head shape are "
        "tests/golden/metric_shapes.npz (N = 200, 4 x 128)")
    if telem is not None:
        res["telemetry"] = {"headline": telem.summary(),
                            "when": "sampled during the warm-up steps in front of each timed loop and an untimed replay of the same steps behind it, never inside a timed region"}
    attach_traffic(res, args, att)
    if args.workload == "M-full" and not args.no_extra:
        res["extra_workloads"] = extra_workloads(args, rank, world, device)
    if rank == 0:
        try:
            res["calibration"] = calibration(device)
            res["measured_copy_GBps"] = res["calibration"]["copy_GBps"]
            c = res["calibration"]
            # one-number box normalisers: the attention step against what this box's read stream would need for the same
            # algorithmic bytes, and (layer section below) the fused projection against this box's MFMA stream
            c["attn_fwd_bwd_bytes_over_read_stream"] = round(res["roofline_fwd_bwd"]["achieved"] / c["read_stream_GBps"], 3)
        except Exception as e:  # pragma: no cover
            res["calibration"] = {"error": repr(e)[:300]}
    if dist.is_initialized():
        try:
            res["rccl"] = rccl_section(world, device)
            res["rccl"]["job"] = dp.describe_ranks(device)
        except Exception as e:  # pragma: no cover
            res["rccl"] = {"error": repr(e)[:300]}
    if not args.no_layer:
        try:
            telem_l = Telemetry(device) if rank == 0 else None
            res["layer"] = layer_section(args, rank, world, device, telem_l)
            if telem_l is not None:
                res.setdefault("telemetry", {})["layer"] = telem_l.summary()
            fused = res["layer"].get("projections", {}).get("uvqk_fwd_fused")
            if rank == 0 and fused and isinstance(res.get("calibration"), dict) and "mfma_stream_tflops" in res["calibration"]:
                res["calibration"]["uvqk_fwd_fused_over_mfma_stream"] = round(fused["tflops"] / res["calibration"]["mfma_stream_tflops"], 3)
        except Exception as e:  # the headline number must survive a failure of the secondary section
            res["layer"] = {"error": repr(e)[:300]}
    if rank == 0 and cpu_res is not None:
        res["cpu_baseline"] = cpu_res
    if rank == 0:
        out.emit(json.dumps(res))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawned(local_rank, args, port):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(args.gpus),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    run(args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="M-full", choices=sorted(WORKLOADS))
    ap.add_argument("--users-per-gpu", type=int, default=None)
    ap.add_argument("--max-seq-len", type=int, default=None)
    ap.add_argument("--heads", type=int, default=None)
    ap.add_argument("--head-dim", type=int, default=None)
    ap.add_argument("--layer-users-per-gpu", type=int, default=1024)
    ap.add_argument("--layer-steps", type=int, default=10)
    ap.add_argument("--layer-dropout", type=float, default=0.1, help="output_dropout_ratio of the layer section (DLRM-v3: 0.1)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other workloads (extra_workloads)")
    ap.add_argument("--layer-tunableop", action="store_true",
                    help="let PyTorch's TunableOp pick the hipBLASLt / rocBLAS solution of the layer section's six GEMM shapes during its "
                         "warm-up (~20 s; measured 15.1 -> 14.5 ms per step, profiles/r03_layer_tunableop.txt); off by default")
    ap.add_argument("--extra-steps", type=int, default=8)
    ap.add_argument("--prewarm-s", type=float, default=0.3, help="seconds of untimed steps in front of every section's warm-up steps (clock / power ramp)")
    ap.add_argument("--parity-users", type=int, default=32, help="users of the timed batch checked against the oracle after the timed loop (0: off)")
    ap.add_argument("--cpu-users", type=int, default=128)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--sort-by-length", type=int, default=None, help="1: heavy-first workgroup order (default: on for C3)")
    ap.add_argument("--no-layer", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--selftest-dist", action="store_true", help="CPU / gloo run of the N-rank scaffolding only (tests)")
    args = ap.parse_args()
    n, h, d, users, _ = WORKLOADS[args.workload]
    args.max_seq_len = args.max_seq_len or n
    args.heads = args.heads or h
    args.head_dim = args.head_dim or d
    args.users_per_gpu = args.users_per_gpu or users
    args.sort_by_length = bool(args.sort_by_length) if args.sort_by_length is not None else args.workload == "C3"
    if args.workload == "C5":
        args.no_layer = True       # the long-history workload has its own memory budget
    if args.workload.startswith("C"):   # the secondary sections belong to the metric shape
        args.no_layer = args.no_cpu = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks here (reference: main.py:68-78 mp.start_processes)
        import torch.multiprocessing as mp

        mp.spawn(_spawned, args=(args, _free_port()), nprocs=args.gpus, join=True)
        return
    run(args)


if __name__ == "__main__":
    main()
