"""CPU tests of bench.py's host-side helpers (no kernels): the telemetry sampler without hwmon nodes, the oracle check of a
sample of the timed batch (fed with the oracle's own results, so the arithmetic of the check is what is tested), the committed
PMC traffic file the line attaches, and the workload table."""

import importlib.util
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_helpers_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_helpers_under_test"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_telemetry_without_hwmon_nodes_reports_its_source_and_never_raises():
    bench = _bench()
    t = bench.Telemetry(torch.device("cpu"))
    t.start()
    t.stop()
    with t:
        pass
    s = t.summary()
    assert "source" in s
    json.dumps(s)                       # goes into the bench line


def test_parity_sample_arithmetic_on_the_oracles_own_results():
    from oracle import hstu_oracle as O

    bench = _bench()
    g = torch.Generator().manual_seed(5)
    B, H, d, N = 7, 2, 16, 24
    lengths = torch.tensor([24, 3, 17, 24, 1, 9, 12])
    off = torch.zeros(B + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lengths, 0)
    L = int(off[-1])
    q, k, v = (torch.empty(L, H, d).uniform_(-0.5, 0.5, generator=g).bfloat16() for _ in range(3))
    do = torch.randn(L, H, d, generator=g).bfloat16()
    alpha = d ** -0.5
    f = lambda t: t.double().numpy()
    out = torch.from_numpy(O.hstu_mha_fwd(N, alpha, f(q), f(k), f(v), off.numpy())).bfloat16()
    dq, dk, dv = (torch.from_numpy(a).bfloat16() for a in O.hstu_mha_bwd(N, alpha, f(do), f(q), f(k), f(v), off.numpy()))
    res = bench.parity_sample(q, k, v, do, out, dq, dk, dv, off, None, N, alpha, n_users=4)
    assert res["users_checked"] == 4 and res["ok"] and set(res["rel_fro"]) == {"out", "dq", "dk", "dv"}
    assert 0 < res["max_rel_fro"] < 3.8e-3          # one bf16 rounding of the exact result
    bad = bench.parity_sample(q, k, v, do, out, dq, dk * 1.05, dv, off, None, N, alpha, n_users=B)
    assert not bad["ok"] and bad["rel_fro"]["dk"] > 3.8e-3 and bad["users_checked"] == B


def test_traffic_entries_are_keyed_on_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py attaches committed PMC traffic only when the entry was taken on the kernel sources it runs on: an entry
    stamped with the current hash is used, one with another stamp (or none) comes back as stale and is refused"""
    bench = _bench()
    cur = bench.kernel_sources_sha256()
    assert len(cur) == 64 and cur == bench.kernel_sources_sha256()
    prof = tmp_path / "profiles"
    prof.mkdir()
    src = {"workload": "M-full", "users_per_gpu": 8192, "head_dim": 128, "heads": 4, "dtype": "bf16"}
    side = lambda k, b: {"kernel": k, "hbm_bytes_per_launch": b}
    fresh = {"source": src, "sources_sha256": cur, "fwd": side("f", 6.7e9), "bwd": side("b", 11.7e9)}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_sources_sha256", lambda: cur)
    json.dump({"entries": [dict(fresh, sources_sha256="0" * 64)]}, open(prof / bench.TRAFFIC_FILES[0], "w"))
    name, ent = bench.traffic_entry("M-full", 8192, 128, 4)
    assert name == bench.TRAFFIC_FILES[0] and "stale" in ent and "changed" in ent["stale"]["why"]
    res = {"roofline": {"kernel": "b", "algorithmic_bytes_per_launch": 11.744e9}, "roofline_fwd": {}}
    args = type("A", (), dict(workload="M-full", users_per_gpu=8192, head_dim=128, heads=4))
    bench.attach_traffic(res, args, None)
    assert res["roofline"]["traffic"] is None and "traffic_stale" in res["roofline"]
    json.dump({"entries": [{k: v for k, v in fresh.items() if k != "sources_sha256"}]}, open(prof / bench.TRAFFIC_FILES[0], "w"))
    assert "no hash" in bench.traffic_entry("M-full", 8192, 128, 4)[1]["stale"]["why"]
    json.dump({"entries": [fresh]}, open(prof / bench.TRAFFIC_FILES[0], "w"))
    name, ent = bench.traffic_entry("M-full", 8192, 128, 4)
    assert ent is fresh or ent == fresh
    res = {"roofline": {"kernel": "b", "algorithmic_bytes_per_launch": 11.744e9}, "roofline_fwd": {}}
    bench.attach_traffic(res, args, None)
    assert res["roofline"]["traffic"] == 11.7e9 and res["roofline_fwd"]["traffic"] == 6.7e9
    assert bench.traffic_entry("C3", 8192, 16, 4) == (None, None)


def test_committed_traffic_entries_match_the_algorithmic_bytes_of_the_metric_shape():
    """whatever committed entry of the metric shape there is -- fresh or stale -- was measured at ~1.00x the algorithmic bytes"""
    bench = _bench()
    found = 0
    for name in bench.TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        for ent in json.load(open(path)).get("entries", []):
            src = ent.get("source", {})
            if (src.get("workload"), src.get("users_per_gpu"), src.get("head_dim"), src.get("heads")) != ("M-full", 8192, 128, 4):
                continue
            found += 1
            rows = 8192 * 200
            for side, per_token in (("fwd", 4 * 4 * 128 * 2), ("bwd", 7 * 4 * 128 * 2)):
                ratio = ent[side]["hbm_bytes_per_launch"] / (rows * per_token)
                assert 0.95 < ratio < 1.05, (name, side, ratio)
    assert found >= 1


def test_workload_table_and_lengths():
    bench = _bench()
    gen = torch.Generator().manual_seed(0)
    for wl, (n, h, d, users, _) in bench.WORKLOADS.items():
        ln = bench.make_lengths(wl if wl not in ("M-targets", "C4", "C5", "L2048") else "M-jag", 64, n, gen, "cpu")
        assert ln.shape == (64,) and int(ln.max()) <= n and int(ln.min()) >= 0
    assert float(np.mean(bench.make_lengths("M-full", 8, 200, gen, "cpu").numpy())) == 200.0
