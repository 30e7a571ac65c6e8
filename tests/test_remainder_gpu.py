"""The folded backward's remainder step (csrc/hstu_attn_bwd_fold.cuh, fold_dq_mini): a user whose last tile holds at most 8 rows
-- the metric shape, L = 200 = 6 x 32 + 8 -- runs that tile as a step of its own in front of the folded loop.

Checked against the fp64 oracle (ops path: hstu_mha forward + backward, 16-bit I/O, head dims 128 and 64) on batches that mix
remainder users (32 k + 1 .. 32 k + 8 rows, every k the kernel takes) with their neighbours on the other side of the boundary
(32 k, 32 k + 9 .. 32 k + 12), long enough for the persistent workgroups to walk from one kind of problem to the other (the next
problem's K/V tiles are requested from the previous problem's tail), with and without target rows.  Reference semantics:
ops/pytorch/pt_hstu_attention.py:87-168 (restated by oracle/hstu_oracle.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
# the suite's 16-bit gates for tensors of this size (tests/test_metric_shapes_gpu.py): 1.5 x the measured error
GATE = {torch.bfloat16: 3.6e-3, torch.float16: 4.5e-4}


def _case(seed, N, B, H, d, dtype, lengths, targets):
    from generative_recommenders_amd.ops.hstu_attention import hstu_mha
    from oracle import hstu_oracle as O

    rng = np.random.default_rng(seed)
    off = O.complete_cumsum(lengths.astype(np.int64))
    L = int(off[-1])
    mk = lambda: torch.from_numpy(rng.standard_normal((L, H, d)) * 0.4).to(dtype)  # noqa: E731
    q, k, v = mk(), mk(), mk()
    g = torch.from_numpy(rng.standard_normal((L, H, d))).to(dtype)
    kw = {}
    if targets:
        kw["num_targets"] = np.minimum(rng.integers(0, 9, size=B), lengths)
    qd, kd, vd = (t.to(DEV).requires_grad_() for t in (q, k, v))
    tkw = {n: torch.from_numpy(x.astype(np.int64)).to(DEV) for n, x in kw.items()}
    alpha = d ** -0.5
    out = hstu_mha(N, alpha, qd, kd, vd, torch.from_numpy(off).to(DEV), **tkw)
    out.backward(g.to(DEV))
    torch.cuda.synchronize()
    q6, k6, v6, g6 = (t.double().numpy() for t in (q, k, v, g))
    ref = O.hstu_mha_fwd(N, alpha, q6, k6, v6, off, **kw)
    rq, rk, rv = O.hstu_mha_bwd(N, alpha, g6, q6, k6, v6, off, **kw)
    bad = []
    for name, got, want in (("out", out, ref), ("dq", qd.grad, rq), ("dk", kd.grad, rk), ("dv", vd.grad, rv)):
        gnp = got.detach().double().cpu().numpy()
        assert np.isfinite(gnp).all(), name
        rel = np.linalg.norm(gnp - want) / np.linalg.norm(want)
        if rel > GATE[dtype]:
            bad.append((name, rel))
        # per user too: a wrong remainder is 8 rows of 200 and could hide in the batch norm
        for b in range(B):
            lo, hi = int(off[b]), int(off[b + 1])
            if hi - lo < 2:
                continue
            den = np.linalg.norm(want[lo:hi])
            r = np.linalg.norm(gnp[lo:hi] - want[lo:hi]) / den if den > 0 else 0.0
            # (a single short user is a handful of roundings: the per-user gate is the format's half ulp x 1.5, as tools/fuzz_attention.py)
            if r > 1.5 * (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11):
                bad.append((f"{name}[user {b}, {hi - lo} rows]", r))
    assert not bad, bad[:8]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("targets", [False, True])
def test_remainder_users_among_their_neighbours(d, dtype, targets):
    """every tile count 2..7, remainders 1..8 next to 0 (= a full last tile) and 9..12, 700 users x 2 heads on 256 CUs"""
    rng = np.random.default_rng(77 + d + int(targets))
    B, H, N = 700, 2, 224
    kt = rng.integers(1, 7, size=B)                      # full tiles in front of the last one
    r = rng.choice([0, 1, 2, 3, 5, 7, 8, 8, 8, 9, 12, 31], size=B)
    lengths = np.minimum(32 * kt + r, N)
    lengths[:4] = [200, 200, 40, 33]
    lengths[rng.random(B) < 0.03] = 0
    _case(1, N, B, H, d, dtype, lengths, targets)


@pytest.mark.parametrize("N", [40, 72, 104, 136, 168, 200])
def test_all_users_at_a_remainder_length(N):
    """max_seq_len itself a remainder length, every user full (dist. of the metric workload), 4 heads of 128, bf16"""
    B = 300
    _case(N, N, B, 4, 128, torch.bfloat16, np.full(B, N), False)


def test_remainder_of_fewer_than_eight_rows_every_count():
    """remainders 1..8 at 6 full tiles (193..200 rows), and the shortest schedule (33..40 rows: one full tile)"""
    lengths = np.array(list(range(193, 201)) * 20 + list(range(33, 41)) * 20)
    _case(5, 200, len(lengths), 3, 128, torch.bfloat16, lengths, False)
