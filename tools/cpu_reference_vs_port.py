#!/usr/bin/env python3
"""The reference's OWN CPU path next to bench.py's port of it, on the same inputs and cores (build container only).

SURVEY 8(d) asks for the reference's ``hstu_mha(kernel=HammerKernel.PYTORCH)`` timed on the GPU box's host cores.  The reference is
Python and /root/reference does not exist on the GPU box -- by the rules of this build it may not travel there in any form -- so
what bench.py times there is oracle/dense_torch.py (``cpu_baseline.kind = "port"``), a restatement of the same padded-dense algorithm
that tests/test_oracle_golden.py pins to the reference's outputs.  This tool closes the gap where the reference DOES exist: it
imports the unmodified reference (with tests/golden/_fbgemm_shim.py for the three absent fbgemm ops), runs both on the SAME
seeded sample of the metric workload, interleaved, checks that the results agree, and writes the ratio of the two timings to
profiles/r06_cpu_reference_vs_port.json.  bench.py quotes that ratio beside its port timing (and, when /root/reference is
importable -- i.e. here -- times the reference itself: ``kind = "reference"``).

    python tools/cpu_reference_vs_port.py [--users 128] [--threads 8] [--passes 7]
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=128)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--passes", type=int, default=7)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_cpu_reference_vs_port.json"))
    args = ap.parse_args()
    import _fbgemm_shim  # noqa: F401
    from generative_recommenders.common import HammerKernel
    from generative_recommenders.ops.hstu_attention import hstu_mha

    from oracle.dense_torch import dense_hstu_mha

    torch.set_num_threads(args.threads)
    res = {}
    for wl, lo in (("M-full", 200), ("M-jag", 180)):
        N, H, d, B = 200, 4, 128, args.users
        gen = torch.Generator().manual_seed(1001)
        lengths = torch.full((B,), N, dtype=torch.int64) if wl == "M-full" else torch.randint(lo, N, (B,), generator=gen)
        off = torch.zeros(B + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(lengths, 0)
        L = int(off[-1])
        q, k, v = (torch.empty(L, H, d).uniform_(-0.01, 0.01, generator=gen).requires_grad_() for _ in range(3))
        do = torch.randn(L, H, d, generator=gen)
        alpha = d ** -0.5

        def run_ref():
            out = hstu_mha(max_seq_len=N, alpha=alpha, q=q, k=k, v=v, seq_offsets=off, causal=True, dropout_pr=0.0, training=True,
                           kernel=HammerKernel.PYTORCH)
            out.backward(do)
            g = (out.detach().clone(), q.grad.clone(), k.grad.clone(), v.grad.clone())
            q.grad = k.grad = v.grad = None
            return g

        def run_port():
            out = dense_hstu_mha(N, alpha, q, k, v, off)
            out.backward(do)
            g = (out.detach().clone(), q.grad.clone(), k.grad.clone(), v.grad.clone())
            q.grad = k.grad = v.grad = None
            return g

        a, b = run_ref(), run_port()          # warm-up + agreement
        diff = max(float((x - y).abs().max() / max(float(y.abs().max()), 1e-30)) for x, y in zip(a, b))
        tr, tp = [], []
        for _ in range(args.passes):
            t0 = time.perf_counter(); run_ref(); tr.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); run_port(); tp.append(time.perf_counter() - t0)
        mr, mp = statistics.median(tr), statistics.median(tp)
        res[wl] = {"users": B, "reference_user_seqs_per_s": B / mr, "port_user_seqs_per_s": B / mp, "port_over_reference": mr / mp,
                   "reference_ms": mr * 1e3, "port_ms": mp * 1e3, "max_rel_diff_of_results": diff, "passes": args.passes}
        print(wl, json.dumps(res[wl]))
    out = {"what": "reference hstu_mha(kernel=HammerKernel.PYTORCH) (imported unmodified from /root/reference, fbgemm ops through "
                   "tests/golden/_fbgemm_shim.py) against oracle/dense_torch.py::dense_hstu_mha, fp32, fwd+bwd, same inputs, interleaved passes, "
                   "medians", "threads": args.threads, "host": "build container (no GPU)", "torch": torch.__version__, "workloads": res}
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
