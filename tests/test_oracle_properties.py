"""The oracle's hand-derived backward passes against central finite differences of its own forward (fp64), and the
reference's metamorphic properties restated on the oracle.  Independent of the golden fixtures: a second pin on the
formulas every GPU parity test is measured against.  CPU only, small sizes."""
import numpy as np
import pytest

from oracle import hstu_oracle as O


def _fd(fn, x, g, eps=1e-6):
    """d <fn(x), g> / dx by central differences (x is perturbed in place and restored)."""
    out = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        old = x[i]
        x[i] = old + eps
        fp = (fn() * g).sum()
        x[i] = old - eps
        fm = (fn() * g).sum()
        x[i] = old
        out[i] = (fp - fm) / (2 * eps)
    return out


@pytest.mark.parametrize("targets,window,ctx,full", [(False, 0, 0, 0), (True, 0, 0, 0), (True, 3, 0, 0), (True, 3, 2, 4), (False, 2, 3, 0)])
def test_attention_backward_matches_finite_differences(targets, window, ctx, full):
    rng = np.random.default_rng(1)
    lengths = np.array([7, 0, 11, 4]) + ctx
    off = O.complete_cumsum(lengths.astype(np.int64))
    L, H, dqk, dv, N = int(off[-1]), 2, 5, 3, int(lengths.max())
    q, k = rng.standard_normal((L, H, dqk)), rng.standard_normal((L, H, dqk))
    v, g = rng.standard_normal((L, H, dv)), rng.standard_normal((L, H, dv))
    nt = np.array([2, 0, 3, 1]) if targets else None
    kw = dict(num_targets=nt, max_attn_len=window, contextual_seq_len=ctx, min_full_attn_seq_len=full)
    fwd = lambda: O.hstu_mha_fwd(N, 0.4, q, k, v, off, **kw)
    dq, dk, dvv = O.hstu_mha_bwd(N, 0.4, g, q, k, v, off, **kw)
    np.testing.assert_allclose(dq, _fd(fwd, q, g), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(dk, _fd(fwd, k, g), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(dvv, _fd(fwd, v, g), rtol=1e-6, atol=1e-8)


def test_rel_bias_attention_backward_matches_finite_differences():
    rng = np.random.default_rng(2)
    n, B, H, d = 9, 3, 2, 4
    lengths = np.array([9, 4, 6])
    off = O.complete_cumsum(lengths.astype(np.int64))
    L = int(off[-1])
    ts = np.sort(rng.integers(0, 10**6, size=(B, n)), axis=1).astype(np.int64)
    q, k, v, g = (rng.standard_normal((L, H, d)) for _ in range(4))
    pos_w, ts_w = rng.standard_normal(2 * n - 1) * 0.3, rng.standard_normal(129) * 0.3
    fwd = lambda: O.rel_bias_attention_fwd(n, q, k, v, off, ts, pos_w, ts_w)
    dq, dk, dv, dpos, dts = O.rel_bias_attention_bwd(n, g, q, k, v, off, ts, pos_w, ts_w)
    for got, x in ((dq, q), (dk, k), (dv, v), (dpos, pos_w), (dts, ts_w)):
        np.testing.assert_allclose(got, _fd(fwd, x, g), rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("l2,table_l2", [(True, True), (False, False), (True, False)])
def test_sampled_softmax_backward_matches_finite_differences(l2, table_l2):
    rng = np.random.default_rng(3)
    n, R, D, V = 5, 6, 4, 9
    q, pos, table = rng.standard_normal((n, D)), rng.standard_normal((n, D)), rng.standard_normal((V, D))
    pos_ids = rng.integers(0, V, size=n)
    rows = rng.integers(0, V, size=(n, R))
    rows[:, 0] = pos_ids                                   # one masked (-5e4) logit per row
    w = rng.random(n) + 0.1
    args = (pos_ids, rows, rows, table, w, 0.7, l2)
    loss = lambda: np.asarray(O.sampled_softmax_fwd(q, pos, *args, table_l2_norm=table_l2)[0])
    dq, dpos, dtable = O.sampled_softmax_bwd(q, pos, *args, table_l2_norm=table_l2)
    one = np.asarray(1.0)
    np.testing.assert_allclose(dq, _fd(loss, q, one), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(dpos, _fd(loss, pos, one), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(dtable, _fd(loss, table, one), rtol=1e-5, atol=1e-8)


def test_norm_and_position_backward_match_finite_differences():
    rng = np.random.default_rng(4)
    x, g = rng.standard_normal((6, 8)), rng.standard_normal((6, 8))
    np.testing.assert_allclose(O.l2_norm_bwd(g, x), _fd(lambda: O.l2_norm_fwd(x), x, g), rtol=1e-6, atol=1e-8)
    w = rng.standard_normal(8)
    dx = O.layer_norm_bwd(g, x, w, 1e-5)[0]
    np.testing.assert_allclose(dx, _fd(lambda: O.layer_norm_fwd(x, w, np.zeros(8), 1e-5), x, g), rtol=1e-6, atol=1e-8)


def test_delta_attention_is_the_tail_of_full_attention():
    """ops/tests/hstu_attention_test.py:356-486, on the oracle itself."""
    rng = np.random.default_rng(5)
    B, H, d, delta = 4, 2, 6, 3
    lengths = rng.integers(delta, 12, size=B)
    off = O.complete_cumsum(lengths.astype(np.int64))
    L, N = int(off[-1]), int(lengths.max())
    q, k, v = (rng.standard_normal((L, H, d)) for _ in range(3))
    nt = rng.integers(1, delta + 1, size=B)
    full = O.hstu_mha_fwd(N, 0.3, q, k, v, off, num_targets=nt, max_attn_len=4, contextual_seq_len=2)
    idx = np.concatenate([np.arange(off[b + 1] - delta, off[b + 1]) for b in range(B)])
    dl = O.delta_hstu_mha_fwd(N, 0.3, q[idx], k, v, off, num_targets=nt, max_attn_len=4, contextual_seq_len=2)
    np.testing.assert_allclose(dl, full[idx], rtol=1e-12, atol=1e-14)


def test_target_rows_do_not_see_each_other():
    """modules/tests/stu_test.py:184-323: with num_targets, swapping two target rows of a user swaps their outputs and
    leaves every other row alone (targets attend to the history and to themselves only)."""
    rng = np.random.default_rng(6)
    L, H, d = 10, 2, 4
    off = np.array([0, L], dtype=np.int64)
    q, k, v = (rng.standard_normal((L, H, d)) for _ in range(3))
    nt = np.array([3])
    out = O.hstu_mha_fwd(L, 0.5, q, k, v, off, num_targets=nt)
    perm = np.arange(L)
    perm[[L - 1, L - 3]] = perm[[L - 3, L - 1]]
    out_p = O.hstu_mha_fwd(L, 0.5, q[perm], k[perm], v[perm], off, num_targets=nt)
    np.testing.assert_allclose(out_p, out[perm], rtol=1e-12, atol=1e-14)
