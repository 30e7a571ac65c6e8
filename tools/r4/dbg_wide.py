#!/usr/bin/env python3
"""debug: where is the wide backward wrong?  per (user, head, tile) relative error of dq / dk / dv vs the oracle"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from generative_recommenders_amd.ops.hstu_attention import hstu_mha
from oracle import hstu_oracle as O

def run(lengths, H, N, alpha, seed, std, targets=None):
    rng = np.random.default_rng(seed)
    off = O.complete_cumsum(np.asarray(lengths, dtype=np.int64)); L = int(off[-1]); d = 128
    mk = lambda: torch.from_numpy(rng.standard_normal((L, H, d)) * std).to(torch.bfloat16)
    q, k, v = mk(), mk(), mk()
    g = torch.from_numpy(rng.standard_normal((L, H, d))).to(torch.bfloat16)
    qd, kd, vd = (t.cuda().requires_grad_() for t in (q, k, v))
    kw = {}
    if targets is not None: kw["num_targets"] = torch.from_numpy(np.asarray(targets, dtype=np.int64)).cuda()
    out = hstu_mha(N, alpha, qd, kd, vd, torch.from_numpy(off).cuda(), **kw)
    out.backward(g.cuda()); torch.cuda.synchronize()
    okw = {} if targets is None else {"num_targets": np.asarray(targets, dtype=np.int64)}
    rq, rk, rv = O.hstu_mha_bwd(N, alpha, g.double().numpy(), q.double().numpy(), k.double().numpy(), v.double().numpy(), off, **okw)
    for name, got, want in (("dq", qd.grad, rq), ("dk", kd.grad, rk), ("dv", vd.grad, rv)):
        gnp = got.double().cpu().numpy()
        bad = []
        for b in range(len(lengths)):
            for h in range(H):
                for t in range((lengths[b] + 31) // 32):
                    r0, r1 = off[b] + 32 * t, min(off[b] + 32 * t + 32, off[b + 1])
                    for db in range(4):
                        w, x = want[r0:r1, h, 32 * db:32 * db + 32], gnp[r0:r1, h, 32 * db:32 * db + 32]
                        rel = np.linalg.norm(x - w) / max(np.linalg.norm(w), 1e-30)
                        if rel > 0.02: bad.append((b, h, t, db, round(rel, 3)))
        print(f"  {name}: total rel {np.linalg.norm(gnp - want) / np.linalg.norm(want):.3e}; bad blocks {len(bad)}: {bad[:6]}")

for cfg in [dict(lengths=[0, 85, 87, 89], H=3, N=89, alpha=0.0884, seed=1, std=0.4),
            dict(lengths=[85, 87, 89], H=3, N=224, alpha=0.0884, seed=1, std=0.4),
            dict(lengths=[95, 96, 97], H=2, N=224, alpha=0.0884, seed=1, std=1.0),
            dict(lengths=[185] * 5, H=4, N=185, alpha=0.37, seed=2, std=0.4, targets=[0, 28, 22, 30, 18]),
            dict(lengths=[200] * 3, H=4, N=200, alpha=0.0884, seed=3, std=0.4)]:
    print(cfg)
    run(**cfg)
