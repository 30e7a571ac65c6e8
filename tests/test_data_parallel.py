"""CPU tests of the multi-GPU path: world_size-2 gloo process group, user sharding and the
bucketed gradient all-reduce (the N>1 path of bench.py; RCCL on the GPU box)."""

import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from generative_recommenders_amd import data_parallel as dp

    r, lr, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)  # identical replicas
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4))
    lengths = torch.tensor([5, 1, 9, 3, 7, 2, 8, 4])
    mine = dp.shard_users(lengths, rank, world)
    g = torch.Generator().manual_seed(123)
    data = torch.randn(8, 8, generator=g)
    loss = model(data[mine]).pow(2).sum()
    loss.backward()
    red = dp.GradientAllReducer(model.parameters(), bucket_bytes=256, average=False)
    assert len(red.buckets) > 1
    red.reduce()
    # reference: single-process gradient over the whole batch
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 4))
    ref(data).pow(2).sum().backward()
    for p, q in zip(model.parameters(), ref.parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-5, atol=1e-6)
    assert dp.max_over_ranks(float(rank)) == float(world - 1)
    assert dp.sum_over_ranks(1.0) == float(world)
    dist.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_gloo_world2_shard_and_allreduce(tmp_path):
    port = 29600 + (os.getpid() % 200)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _worker_overlap(rank, world, port, out_dir):
    """per-layer buckets reduced from inside backward (hooks) == the single-process gradient; two steps, with p.grad reset
    to None in between as a training loop does"""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from generative_recommenders_amd import data_parallel as dp

    dp.init_from_env(backend="gloo")
    mk = lambda: torch.nn.Sequential(*[torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Tanh()) for _ in range(3)])
    torch.manual_seed(0)
    model = mk()
    red = dp.GradientAllReducer(None, buckets=[layer.parameters() for layer in model], overlap=True, average=False, check_every=1)
    assert len(red.buckets) == 3
    lengths = torch.tensor([5, 1, 9, 3, 7, 2, 8, 4])
    mine = dp.shard_users(lengths, rank, world)
    g = torch.Generator().manual_seed(123)
    torch.manual_seed(0)
    ref = mk()
    for step in range(2):
        data = torch.randn(8, 8, generator=g)
        for p in model.parameters():
            p.grad = None
        model(data[mine]).pow(2).sum().backward()      # the collectives start inside this call
        red.reduce()
        for p in ref.parameters():
            p.grad = None
        ref(data).pow(2).sum().backward()
        for p, q in zip(model.parameters(), ref.parameters()):
            torch.testing.assert_close(p.grad, q.grad, rtol=1e-5, atol=1e-6)
    # gradient accumulation: two micro-batches, the first under no_sync() -- one collective per bucket, on the sums
    data = torch.randn(8, 8, generator=g)
    for p in model.parameters():
        p.grad = None
    half = len(mine) // 2
    with red.no_sync():
        model(data[mine[:half]]).pow(2).sum().backward()
    model(data[mine[half:]]).pow(2).sum().backward()
    red.reduce()
    for p in ref.parameters():
        p.grad = None
    ref(data).pow(2).sum().backward()
    for p, q in zip(model.parameters(), ref.parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-5, atol=1e-6)
    # a second backward without reduce() / no_sync() is refused instead of reducing half a gradient
    for p in model.parameters():
        p.grad = None
    model(data[mine]).pow(2).sum().backward()
    try:
        model(data[mine]).pow(2).sum().backward()
        raise AssertionError("expected the reducer to refuse a second backward before reduce()")
    except RuntimeError as e:
        assert "no_sync" in str(e)
    red.reduce()
    info = dp.describe_ranks()
    assert info["backend"] == "gloo" and [r["rank"] for r in info["ranks"]] == list(range(world))
    dist.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def _worker_mixed(rank, world, port, out_dir):
    """an explicit bucket of mixed dtypes (fp32 / bf16 / fp32) is split by dtype: every gradient is reduced (round 3's reducer
    re-allocated the flat buffer at the first bf16 gradient and left the fp32 ones out of the collective); a bucket with a
    parameter that gets no gradient is reduced in reduce(), on every rank alike"""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from generative_recommenders_amd import data_parallel as dp

    dp.init_from_env(backend="gloo")
    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.randn(4, 3))
    b = torch.nn.Parameter(torch.randn(3, 5).to(torch.bfloat16))
    c = torch.nn.Parameter(torch.randn(5))
    unused = torch.nn.Parameter(torch.randn(2))
    red = dp.GradientAllReducer(None, buckets=[[a, b, c], [unused, torch.nn.Parameter(torch.randn(2))]], overlap=True, average=False, check_every=1)
    assert [len(bk) for bk in red.buckets] == [2, 1, 2] and red.buckets[1][0] is b
    used2 = red.buckets[2][1]
    x = torch.full((2, 4), float(rank + 1))
    y = ((x @ a).to(torch.bfloat16) @ b).float() + c
    (y.sum() + used2.sum() * (rank + 1)).backward()
    red.reduce()
    # the same on one process with both ranks' inputs
    tot = {}
    for r in range(world):
        aa, bb, cc = (t.detach().clone().requires_grad_() for t in (a, b, c))
        xx = torch.full((2, 4), float(r + 1))
        (((xx @ aa).to(torch.bfloat16) @ bb).float() + cc).sum().backward()
        for n, t in (("a", aa), ("b", bb), ("c", cc)):
            tot[n] = t.grad.float() + tot.get(n, 0)
    torch.testing.assert_close(a.grad, tot["a"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(b.grad.float(), tot["b"], rtol=2e-2, atol=1e-2)
    torch.testing.assert_close(c.grad, tot["c"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(used2.grad, torch.full((2,), float(sum(range(1, world + 1)))))
    assert unused.grad is not None and float(unused.grad.abs().sum()) == 0.0
    dist.barrier()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_gloo_world2_mixed_dtype_bucket_and_unused_parameter(tmp_path):
    port = 29650 + (os.getpid() % 100)
    mp.spawn(_worker_mixed, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_gloo_world2_overlapped_per_layer_buckets(tmp_path):
    port = 29850 + (os.getpid() % 100)
    mp.spawn(_worker_overlap, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _worker_mispaired(rank, world, port, out_dir):
    """parameter usage that depends on the rank -- the documented precondition violated: rank 0's second layer gets no
    gradient, so its bucket is late there while rank 1 launches it from a hook; the collectives pair bucket 0 with bucket 1
    (equal sizes: nothing fails inside gloo / RCCL).  The bucket tags make BOTH ranks raise in reduce()."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import warnings

    from generative_recommenders_amd import data_parallel as dp

    dp.init_from_env(backend="gloo")
    torch.manual_seed(0)
    l0, l1 = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)
    red = dp.GradientAllReducer(None, buckets=[l0.parameters(), l1.parameters()], overlap=True, average=False)
    x = torch.randn(3, 4)
    y = l0(x) if rank == 0 else l1(l0(x))
    y.sum().backward()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            red.reduce()
            raise AssertionError("expected a bucket tag mismatch")
        except RuntimeError as e:
            assert "tag mismatch" in str(e), str(e)
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_mispaired_buckets_are_detected(tmp_path):
    port = 29750 + (os.getpid() % 100)
    mp.spawn(_worker_mispaired, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _worker_single(rank, world, port, out_dir):
    """a ONE-rank group with single_rank_collectives=True: hooks -> staging copy -> asynchronous all-reduce -> p.grad, and
    no_sync() in between, give bit for bit the gradients of a run without a reducer (the gloo twin of the RCCL test in
    tests/test_rccl_gpu.py)"""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from generative_recommenders_amd import data_parallel as dp

    assert dp.init_from_env(backend="gloo", single_rank_group=True) == (0, 0, 1) and dist.is_initialized()
    mk = lambda: torch.nn.Sequential(*[torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Tanh()) for _ in range(3)])
    torch.manual_seed(0)
    model = mk()
    torch.manual_seed(0)
    ref = mk()
    red = dp.GradientAllReducer(None, buckets=[layer.parameters() for layer in model], overlap=True, single_rank_collectives=True)
    g = torch.Generator().manual_seed(5)
    for step in range(6):
        data = torch.randn(8, 8, generator=g)
        for m in (model, ref):
            for p in m.parameters():
                p.grad = None
        if step % 2:
            with red.no_sync():
                model(data[:4]).pow(2).sum().backward()
            model(data[4:]).pow(2).sum().backward()
            ref(data[:4]).pow(2).sum().backward()
            ref(data[4:]).pow(2).sum().backward()
        else:
            model(data).pow(2).sum().backward()
            ref(data).pow(2).sum().backward()
        red.reduce()
        for p, q in zip(model.parameters(), ref.parameters()):
            assert torch.equal(p.grad, q.grad)
    assert red._calls == 6
    open(os.path.join(out_dir, "ok0"), "w").write("ok")
    dist.destroy_process_group()


def test_gloo_single_rank_group_runs_the_collectives(tmp_path):
    port = 29450 + (os.getpid() % 100)
    mp.spawn(_worker_single, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok0").exists()


def test_nccl_refuses_more_ranks_than_devices(monkeypatch):
    """two ranks on one GPU must fail loudly at start-up, not hang in the first collective"""
    import pytest

    from generative_recommenders_amd import data_parallel as dp

    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(RuntimeError, match="no GPU of its own"):
        dp.init_from_env(backend="nccl")


def test_shard_users_partitions():
    from generative_recommenders_amd import data_parallel as dp

    lengths = torch.randint(1, 200, (37,))
    for mode in ("count", "work"):
        seen = torch.cat([dp.shard_users(lengths, r, 4, mode) for r in range(4)])
        assert sorted(seen.tolist()) == list(range(37))
    # work balancing: L^2 loads within 25 % of each other on a long-tailed distribution
    lengths = torch.cat([torch.full((4,), 200), torch.randint(1, 30, (60,))])
    loads = [float((lengths[dp.shard_users(lengths, r, 4, "work")].double() ** 2).sum()) for r in range(4)]
    assert max(loads) / min(loads) < 1.25
    off = dp.local_offsets(torch.tensor([3, 0, 2]))
    assert off.tolist() == [0, 3, 3, 5]


def test_sharded_attention_equals_the_full_batch():
    """every op on the path is per-user, so a rank's users with locally re-based offsets give exactly the
    rows of the full batch -- checked on the oracle (no forward collective exists to get wrong)."""
    import numpy as np
    from generative_recommenders_amd import data_parallel as dp
    from oracle import hstu_oracle as O

    rng = np.random.default_rng(0)
    lengths = torch.tensor([5, 0, 9, 3, 7, 2, 8, 4, 1])
    off = dp.local_offsets(lengths).numpy()
    L, H, d, N = int(off[-1]), 2, 4, 9
    q, k, v, g = (rng.standard_normal((L, H, d)) for _ in range(4))
    nt = np.minimum(rng.integers(0, 3, size=len(lengths)), lengths.numpy())
    full = O.hstu_mha_fwd(N, 0.5, q, k, v, off, num_targets=nt)
    dq, dk, dv = O.hstu_mha_bwd(N, 0.5, g, q, k, v, off, num_targets=nt)
    for mode in ("count", "work"):
        for rank in range(3):
            mine = dp.shard_users(lengths, rank, 3, mode)
            rows = np.concatenate([np.arange(off[u], off[u + 1]) for u in mine.tolist()] or [np.zeros(0, dtype=np.int64)]).astype(np.int64)
            loc = dp.local_offsets(lengths[mine]).numpy()
            o = O.hstu_mha_fwd(N, 0.5, q[rows], k[rows], v[rows], loc, num_targets=nt[mine.numpy()])
            np.testing.assert_array_equal(o, full[rows])
            sq, sk, sv = O.hstu_mha_bwd(N, 0.5, g[rows], q[rows], k[rows], v[rows], loc, num_targets=nt[mine.numpy()])
            np.testing.assert_array_equal(sq, dq[rows])
            np.testing.assert_array_equal(sk, dk[rows])
            np.testing.assert_array_equal(sv, dv[rows])


def _last_json_line(text):
    import json

    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_spawns_its_own_ranks_and_under_torchrun_on_gloo():
    """`python bench.py --gpus 2` without a launcher spawns its two ranks itself (reference: main.py:68-78); under
    torch.distributed.run it takes them from the environment.  --selftest-dist runs the N-rank scaffolding only (barriers,
    MAX over ranks, the all-reduce probe, ONE JSON line from rank 0) on CPU / gloo."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    base = [os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-dist", "--steps", "3", "--warmup", "1"]
    r = subprocess.run([sys.executable] + base, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["rccl"]["ranks"] == 2 and d["rccl"]["backend"] == "gloo"
    assert d["steps"] == 3 and d["warmup"] == 1 and d["ms_per_step"] > 0 and d["rccl"]["busbw_GBps"] > 0
    assert sum(1 for ln in r.stdout.splitlines() if ln.startswith("{")) == 1          # rank 0 only
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29547"] + base, cwd=root, env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    # a launcher / flag mismatch is an error, not a silent single-rank run
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-dist"], cwd=root,
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
