"""The two-kernel backward for long sequences (csrc/hstu_attn_bwd_long.cuh: dK / dV kernel + dQ kernel; max_seq_len > 224, 16-bit
I/O, no bias, no contextual rows) against the fp64 oracle, through the ops path (hstu_mha forward + backward).

Lengths on both sides of the kernels' block boundaries (7 key tiles per dK / dV block, 128 query rows per dQ block, 32-row tiles),
every mask the path takes (plain causal, target rows, attention window with and without min_full_attn_seq_len), empty and full
users, both head dims and 16-bit dtypes; results must be bit-identical run to run (no atomics anywhere).  Reference semantics:
ops/pytorch/pt_hstu_attention.py:87-168 (restated by oracle/hstu_oracle.py); the kernels replace
ops/triton/triton_hstu_attention.py:899-1764 at the lengths of the reference's own benchmark sweep
(ops/benchmarks/hstu_attention_bench.py:139)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
# per-tensor gates of the suite for 16-bit I/O (1.5 x the measured error; tests/test_metric_shapes_gpu.py)
GATE = {torch.bfloat16: 3.6e-3, torch.float16: 4.5e-4}


def _run(N, H, d, dtype, lengths, seed, **kw):
    from generative_recommenders_amd.ops import _launch
    from generative_recommenders_amd.ops.hstu_attention import hstu_mha
    from oracle import hstu_oracle as O

    rng = np.random.default_rng(seed)
    B = len(lengths)
    lengths = np.asarray(lengths)
    off = O.complete_cumsum(lengths.astype(np.int64))
    L = int(off[-1])
    mk = lambda: torch.from_numpy(rng.standard_normal((L, H, d)) * 0.4).to(dtype)  # noqa: E731
    q, k, v = mk(), mk(), mk()
    g = torch.from_numpy(rng.standard_normal((L, H, d))).to(dtype)
    okw = dict(kw)
    if okw.pop("targets", False):
        okw["num_targets"] = np.minimum(rng.integers(0, 30, size=B), lengths)
    tkw = {n: (torch.from_numpy(x.astype(np.int64)).to(DEV) if isinstance(x, np.ndarray) else x) for n, x in okw.items()}
    alpha = d ** -0.5
    name = _launch.attn_bwd_kernel_name(dtype, d, d, N, heads=H, alpha=alpha, max_attn_len=okw.get("max_attn_len", 0),
                                        contextual_seq_len=okw.get("contextual_seq_len", 0))
    assert name.startswith("hstu_attn_bwd_dkv_kernel"), name
    outs = []
    for _ in range(2):
        qd, kd, vd = (t.to(DEV).requires_grad_() for t in (q, k, v))
        out = hstu_mha(N, alpha, qd, kd, vd, torch.from_numpy(off).to(DEV), **tkw)
        out.backward(g.to(DEV))
        torch.cuda.synchronize()
        outs.append((qd.grad.clone(), kd.grad.clone(), vd.grad.clone()))
    for a, b2 in zip(*outs):
        assert torch.equal(a, b2), "the two-kernel backward must be bit-identical run to run"
    q6, k6, v6, g6 = (t.double().numpy() for t in (q, k, v, g))
    rq, rk, rv = O.hstu_mha_bwd(N, alpha, g6, q6, k6, v6, off, **okw)
    bad = []
    for nm, got, want in (("dq", outs[0][0], rq), ("dk", outs[0][1], rk), ("dv", outs[0][2], rv)):
        gnp = got.double().cpu().numpy()
        assert np.isfinite(gnp).all(), nm
        rel = np.linalg.norm(gnp - want) / np.linalg.norm(want)
        if rel > GATE[dtype]:
            bad.append((nm, rel))
        for b in range(B):      # per user: a wrong block of one user would hide in the batch norm
            lo, hi = int(off[b]), int(off[b + 1])
            den = np.linalg.norm(want[lo:hi])
            if hi - lo >= 8 and den > 0:
                r = np.linalg.norm(gnp[lo:hi] - want[lo:hi]) / den
                if r > 1.5 * GATE[dtype]:
                    bad.append((f"{nm}[user {b}: {hi - lo} rows]", r))
    assert not bad, bad[:8]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("N", [225, 449, 700])
def test_plain_causal(N, d, dtype):
    # full users, users ending right at / behind block boundaries (7 x 32 = 224 keys per dK/dV block, 128 rows per dQ block), short and empty ones
    lengths = [N, 0, 1, 224, min(N, 225), min(N, 257), N - 1, min(N, 448), 31]
    _run(N, 2, d, dtype, lengths, seed=N + d)


@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("mask", ["targets", "window", "window_full", "window_targets", "ctx", "ctx_long", "ctx_window_targets"])
def test_masks(mask, d):
    N = 520
    kw = {"targets": {"targets": True}, "window": {"max_attn_len": 100}, "window_full": {"max_attn_len": 64, "min_full_attn_seq_len": 40},
          "window_targets": {"max_attn_len": 150, "targets": True},
          # contextual rows (id 0: they see every non-target key): their query tiles run in front of every key block's own
          "ctx": {"contextual_seq_len": 5}, "ctx_long": {"contextual_seq_len": 40},
          "ctx_window_targets": {"contextual_seq_len": 7, "max_attn_len": 90, "targets": True}}[mask]
    _run(N, 2, d, torch.bfloat16, [N, 300, 0, 519, 226, 97], seed=17 + d, **kw)


def test_long_rows_1500_many_blocks():
    """7 dK/dV blocks and 12 dQ blocks per user; 3 heads (dispatch groups of 8 (user, head) pairs not full)"""
    _run(1500, 3, 128, torch.bfloat16, [1500, 1463, 700], seed=5)


def test_window_much_shorter_than_the_sequence():
    """the dK/dV kernel stops its query tiles at the window's reach, the dQ kernel starts its key tiles there"""
    _run(1200, 2, 128, torch.bfloat16, [1200, 1000, 333], seed=6, max_attn_len=70)


@pytest.mark.parametrize("N", [100, 200])
def test_contextual_rows_at_short_lengths_take_this_path(N):
    """the folded schedules refuse contextual rows: from 65 rows on such batches run the two kernels (one key block)"""
    _run(N, 2, 128, torch.bfloat16, [N, N - 1, 0, 70, 33, 3], seed=9, contextual_seq_len=6)
    _run(N, 2, 64, torch.float16, [N, 64, 65], seed=10, contextual_seq_len=3, targets=True)
