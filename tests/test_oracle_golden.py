"""Pins the numpy oracle (oracle/hstu_oracle.py) to the golden vectors minted from
the reference's own PyTorch path (tests/golden/make_golden.py).  CPU only."""

import os

import numpy as np
import pytest

from conftest import load_cases, scalar
from oracle import hstu_oracle as O

TOL = dict(rtol=2e-5, atol=2e-6)  # golden is fp32 torch; oracle runs fp64


def _nt(c):
    return c.get("num_targets")


@pytest.mark.parametrize("idx", range(9))
def test_attention_fwd_bwd_matches_reference(idx):
    c = load_cases("attention.npz")[idx]
    kw = dict(
        num_targets=_nt(c), max_attn_len=int(c["max_attn_len"]), contextual_seq_len=int(c["contextual"]),
        min_full_attn_seq_len=int(c["min_full"]),
    )
    N, alpha = int(c["N"]), float(c["alpha"])
    out = O.hstu_mha_fwd(N, alpha, c["q"], c["k"], c["v"], c["offsets"], **kw)
    np.testing.assert_allclose(out, c["out"], **TOL)
    dq, dk, dv = O.hstu_mha_bwd(N, alpha, c["dout"], c["q"], c["k"], c["v"], c["offsets"], **kw)
    np.testing.assert_allclose(dq, c["dq"], **TOL)
    np.testing.assert_allclose(dk, c["dk"], **TOL)
    np.testing.assert_allclose(dv, c["dv_"], **TOL)


@pytest.mark.parametrize("idx", range(3))
def test_delta_attention_matches_reference(idx):
    c = load_cases("delta_attention.npz")[idx]
    out = O.delta_hstu_mha_fwd(
        int(c["N"]), float(c["alpha"]), c["delta_q"], c["k"], c["v"], c["offsets"], num_targets=_nt(c),
        max_attn_len=int(c["max_attn_len"]), contextual_seq_len=int(c["contextual"]),
    )
    np.testing.assert_allclose(out, c["out"], **TOL)


def test_delta_equals_tail_of_full():
    """Metamorphic check of ops/tests/hstu_attention_test.py:356-486."""
    rng = np.random.default_rng(0)
    B, H, d, delta = 3, 2, 16, 4
    lengths = rng.integers(delta, 30, size=B)
    off = O.complete_cumsum(lengths.astype(np.int64))
    N = int(lengths.max())
    q = rng.uniform(-0.1, 0.1, (off[-1], H, d))
    k = rng.uniform(-0.1, 0.1, (off[-1], H, d))
    v = rng.uniform(-0.1, 0.1, (off[-1], H, d))
    nt = rng.integers(1, delta + 1, size=B)
    full = O.hstu_mha_fwd(N, 0.25, q, k, v, off, num_targets=nt)
    dq = np.concatenate([q[off[b + 1] - delta : off[b + 1]] for b in range(B)])
    dl = O.delta_hstu_mha_fwd(N, 0.25, dq, k, v, off, num_targets=nt)
    tail = np.concatenate([full[off[b + 1] - delta : off[b + 1]] for b in range(B)])
    np.testing.assert_allclose(dl, tail, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("idx", range(4))
def test_jagged_concat_split_bit_exact(idx):
    c = load_cases("jagged.npz")[idx]
    da, db = bool(c["dense_a"]), bool(c["dense_b"])
    ma, mb = int(c["ma"]), int(c["mb"])
    cat = O.concat_2D_jagged(c["va"], c["vb"], ma, mb, None if da else c["oa"], None if db else c["ob"])
    assert np.array_equal(cat, c["cat"])
    l, r = O.split_2D_jagged(c["cat"], ma if da else None, mb if db else None,
                             None if da else c["oa"], None if db else c["ob"])
    assert np.array_equal(l, c["split_l"]) and np.array_equal(r, c["split_r"])
    assert np.array_equal(l, c["va"]) and np.array_equal(r, c["vb"])


def test_jagged_l2_prefix_variants_bit_exact():
    c = load_cases("jagged_l2.npz")[0]
    ctx = int(c["ctx"])
    cat = O.concat_2D_jagged(c["px"], c["lx"], None, None, c["op"], c["ol"], n_prefix_from_right=ctx)
    assert np.array_equal(cat, c["cat"])
    p, l = O.split_2D_jagged(c["cat"], None, None, c["op"], c["ol"], n_prefix_to_right=ctx)
    assert np.array_equal(p, c["split_p"]) and np.array_equal(l, c["split_l"])


def test_padded_dense_roundtrip_and_cumsum():
    rng = np.random.default_rng(1)
    lengths = rng.integers(0, 9, size=7).astype(np.int32)
    off = O.complete_cumsum(lengths)
    assert off.dtype == np.int32 and off[0] == 0 and off[-1] == lengths.sum()
    vals = rng.standard_normal((off[-1], 5)).astype(np.float32)
    dense = O.jagged_to_padded_dense(vals, off, 10)
    assert np.array_equal(O.dense_to_jagged(dense, off), vals)
    # truncation when max_len < L
    dense2 = O.jagged_to_padded_dense(vals, off, 3)
    for b in range(7):
        n = min(lengths[b], 3)
        assert np.array_equal(dense2[b, :n], vals[off[b] : off[b] + n])
        assert not dense2[b, n:].any()


def test_1d_jagged_helpers():
    vals = np.array([1, 2, 3, 4, 5, 6], dtype=np.int64)
    off = np.array([0, 2, 2, 6], dtype=np.int64)
    d = O.expand_1d_jagged_to_dense(vals, off, 3)
    assert d.tolist() == [[1, 2, 2], [0, 0, 0], [3, 4, 5]]
    c = O.concat_1d_jagged_jagged(np.array([1, 0, 2]), np.array([7, 8, 9]), np.array([2, 1, 0]), np.array([1, 2, 3]))
    assert c.tolist() == [7, 1, 2, 3, 8, 9]


def _compute(name):
    for c in load_cases("compute.npz"):
        if str(c["name"]) == name:
            return c
    raise KeyError(name)


def test_layer_norm_matches_reference():
    c = _compute("ln")
    y = O.layer_norm_fwd(c["x"], c["w"], c["b"], float(c["eps"]))
    np.testing.assert_allclose(y, c["y"], rtol=2e-5, atol=2e-6)


def test_swish_layer_norm_matches_reference():
    """oracle (fp64) against the reference's pytorch_swish_layer_norm + autograd (fp32): tests/golden/swish_layer_norm.npz"""
    cases = load_cases("swish_layer_norm.npz")
    assert len(cases) == 6
    for c in cases:
        eps = float(c["eps"])
        y = O.swish_layer_norm_fwd(c["x"], c["w"], c["b"], eps)
        np.testing.assert_allclose(y, c["y"], rtol=3e-5, atol=3e-6)
        if "gy" in c:
            dx, dw, db = O.swish_layer_norm_bwd(c["gy"], c["x"], c["w"], c["b"], eps)
            scale = max(1.0, float(np.abs(c["dx"]).max()))
            np.testing.assert_allclose(dx, c["dx"], rtol=2e-4, atol=2e-5 * scale)
            np.testing.assert_allclose(dw, c["dw"], rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(c["dw"]).max())))
            np.testing.assert_allclose(db, c["db"], rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(c["db"]).max())))


def test_uvqk_matches_reference():
    c = _compute("uvqk")
    u, q, k, v = O.hstu_compute_uqvk(c["x"], c["nw"], c["nb"], 1e-6, int(c["H"]), int(c["A"]), int(c["Hd"]),
                                     c["W"], c["beta"])
    for got, want in ((u, c["u"]), (q, c["q"]), (k, c["k"]), (v, c["v"])):
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["out_ln", "out_ln_cat", "out_gn_cat"])
def test_compute_output_matches_reference(name):
    c = _compute(name)
    y = O.hstu_compute_output(c["attn"], c["u"], c["x"], c["nw"], c["nb"], 1e-6, c["Wo"], int(c["H"]),
                              int(c["Ld"]), bool(c["cat"]), bool(c["gn"]))
    np.testing.assert_allclose(y, c["y"], rtol=1e-4, atol=1e-5)


def test_layer_norm_bwd_against_finite_difference():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((5, 12))
    w = rng.standard_normal(12)
    b = rng.standard_normal(12)
    dy = rng.standard_normal((5, 12))
    dx, dw, db = O.layer_norm_bwd(dy, x, w, 1e-6)
    eps = 1e-6
    num = np.zeros_like(x)
    for i in range(5):
        for j in range(12):
            xp = x.copy(); xp[i, j] += eps
            xm = x.copy(); xm[i, j] -= eps
            num[i, j] = ((O.layer_norm_fwd(xp, w, b, 1e-6) - O.layer_norm_fwd(xm, w, b, 1e-6)) * dy).sum() / (2 * eps)
    np.testing.assert_allclose(dx, num, rtol=1e-5, atol=1e-7)


def test_research_rel_bias_attention_matches_reference():
    c = load_cases("research_attention.npz")[0]
    n, H = int(c["n"]), int(c["H"])
    A, Ld = int(c["A"]), int(c["Ld"])
    q = c["q"].reshape(-1, H, A)
    k = c["k"].reshape(-1, H, A)
    v = c["v"].reshape(-1, H, Ld)
    out = O.rel_bias_attention_fwd(n, q, k, v, c["offsets"], c["ts"], c["pos_w"], c["ts_w"])
    np.testing.assert_allclose(out.reshape(c["out"].shape), c["out"], rtol=2e-5, atol=2e-6)
    g = c["g"].reshape(-1, H, Ld)
    dq, dk, dv, dpos, dts = O.rel_bias_attention_bwd(n, g, q, k, v, c["offsets"], c["ts"], c["pos_w"], c["ts_w"])
    np.testing.assert_allclose(dq.reshape(c["dq"].shape), c["dq"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dk.reshape(c["dk"].shape), c["dk"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dv.reshape(c["dv_"].shape), c["dv_"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(dpos, c["dpos_w"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(dts, c["dts_w"], rtol=2e-4, atol=2e-6)


def test_mask_zero_history_and_properties():
    # zero-history user: only targets; each target sees itself only
    m = O.valid_attn_mask(3, 3, num_targets=3)
    assert np.array_equal(m, np.eye(3, dtype=bool))
    # plain causal
    m = O.valid_attn_mask(5, 5)
    assert np.array_equal(m, np.tril(np.ones((5, 5), dtype=bool)))
    # window of 2 without targets: row i sees cols [i-2, i]
    m = O.valid_attn_mask(6, 6, max_attn_len=2)
    for i in range(6):
        for j in range(6):
            assert m[i, j] == (0 <= i - j <= 2)
    # contextual rows see the whole history but not targets
    m = O.valid_attn_mask(8, 8, num_targets=2, contextual_seq_len=3)
    assert m[0, :6].all() and not m[0, 6:].any()
    assert m[7, :6].all() and m[7, 7] and not m[7, 6]


def test_dense_torch_stu_stack_matches_reference():
    """the CPU port bench.py times as the layer baseline (oracle/dense_torch.py::dense_stu_stack) against the reference's
    2-layer STUStack (layer norm + group norm layers, targets), forward and input gradient"""
    import torch

    from oracle.dense_torch import dense_stu_stack

    c = load_cases("stu.npz")[0]
    layers = []
    for li, gn in enumerate((False, True)):
        prm = {k.split(".")[-1]: torch.from_numpy(v) for k, v in c.items() if k.startswith(f"p:_stu_layers.{li}.")}
        layers.append((prm, gn))
    x = torch.from_numpy(c["x"]).requires_grad_()
    y = dense_stu_stack(x, layers, num_heads=int(c["H"]), attn_dim=int(c["A"]), hidden_dim=int(c["Hd"]),
                        max_seq_len=int(c["N"]), seq_offsets=torch.from_numpy(c["offsets"]),
                        num_targets=torch.from_numpy(c["num_targets"]))
    np.testing.assert_allclose(y.detach().numpy(), c["y"], rtol=1e-4, atol=1e-5)
    y.backward(torch.from_numpy(c["gy"]))
    np.testing.assert_allclose(x.grad.numpy(), c["dx"], rtol=1e-4, atol=1e-5)


def _bf16_bits_to_f64(bits):
    import torch

    return torch.from_numpy(np.ascontiguousarray(bits)).view(torch.bfloat16).double().numpy()


@pytest.mark.parametrize("idx", range(2))
def test_oracle_matches_reference_at_the_metric_shapes(idx):
    """M (N = 200, 4 heads of 128) and C2 (N = 211, 4 heads of 64, targets): the shapes the throughput is quoted on
    (tests/golden/make_golden.py::metric_shape_cases).  The oracle runs in fp64, the reference in fp32."""
    c = load_cases("metric_shapes.npz")[idx]
    q, k, v, do = (_bf16_bits_to_f64(c[n]) for n in ("q_bf16", "k_bf16", "v_bf16", "dout_bf16"))
    nt = c.get("num_targets")
    out = O.hstu_mha_fwd(int(c["N"]), float(c["alpha"]), q, k, v, c["offsets"], nt)
    dq, dk, dv = O.hstu_mha_bwd(int(c["N"]), float(c["alpha"]), do, q, k, v, c["offsets"], nt)
    for name, got, want in (("out", out, c["out"]), ("dq", dq, c["dq"]), ("dk", dk, c["dk"]), ("dv", dv, c["dv_"])):
        rel = np.linalg.norm(got - want) / np.linalg.norm(want)
        assert rel < 2e-6, f"{name}: relative Frobenius difference {rel:.2e} (fp32 reference vs fp64 oracle)"
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-5 * np.abs(want).max())


# ------------------------------------------------------------------ §8f rank 1: timestamp / position encoder
@pytest.mark.parametrize("idx", range(3))
def test_oracle_matches_reference_position_encoder(idx):
    c = load_cases("position.npz")[idx]
    nt = c.get("num_targets")
    out, pos_idx, ts_idx = O.add_timestamp_positional_embeddings_fwd(
        float(c["alpha"]), c["x"].astype(np.float64), c["offsets"], c["ts"], c["pos_w"].astype(np.float64),
        c["ts_w"].astype(np.float64), int(c["ctx"]), nt, bool(c["interleave"]), str(c["fn"]), bucket_clamp="pytorch_path")
    np.testing.assert_allclose(out, c["out"], rtol=1e-6, atol=1e-6)
    dx, dpos, dts = O.add_timestamp_positional_embeddings_bwd(float(c["alpha"]), c["g"].astype(np.float64), pos_idx,
                                                               ts_idx, c["pos_w"].shape[0], c["ts_w"].shape[0])
    np.testing.assert_allclose(dx, c["dx"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dpos, c["dpos_w"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dts, c["dts_w"], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ §8f rank 2: output post-processing
def test_oracle_matches_reference_postprocessors():
    c = load_cases("postprocess.npz")[0]
    x, g = c["x"].astype(np.float64), c["g"].astype(np.float64)
    np.testing.assert_allclose(O.l2_norm_fwd(x), c["l2_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(O.l2_norm_bwd(g, x), c["l2_dx"], rtol=2e-5, atol=1e-4)
    y = O.layer_norm_fwd(x, c["ln_w"].astype(np.float64), c["ln_b"].astype(np.float64), 1e-5)
    np.testing.assert_allclose(y, c["ln_out"], rtol=1e-5, atol=1e-5)
    # candidate split + l2 norm (return_full_embeddings = False): postprocessor on the candidate rows only
    cand, idx = O.split_candidates(x, c["lengths"], c["num_targets"])
    np.testing.assert_allclose(O.l2_norm_fwd(cand), c["pp_cand_cand"], rtol=1e-5, atol=1e-6)
    dx = np.zeros_like(x)
    dx[idx] = O.l2_norm_bwd(c["pp_cand_gc"].astype(np.float64), cand)
    np.testing.assert_allclose(dx, c["pp_cand_dx"], rtol=2e-5, atol=1e-4)
    # return_full_embeddings = True: postprocessor on every row, then the split
    full = O.l2_norm_fwd(x)
    np.testing.assert_allclose(full, c["pp_full_emb"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(full[idx], c["pp_full_cand"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ §8f rank 3: sampled-softmax loss
def _ss_inputs(c):
    """(q, pos_emb, pos_ids, neg_rows, neg_ids, table, weights) of a golden sampled-softmax case."""
    kind = str(c["kind"])
    if kind == "local":
        return c["q"], c["pos_emb"], c["pos_ids"], c["sampled_ids"], c["sampled_ids"], c["table"], c["weights"]
    if kind == "in-batch":
        return (c["q"], c["pos_emb"], c["pos_ids"], c["sampled_offsets"], c["cached_ids"][c["sampled_offsets"]],
                c["cached_embeddings"], c["weights"])
    raise AssertionError(kind)


@pytest.mark.parametrize("idx", range(4))
def test_oracle_matches_reference_sampled_softmax(idx):
    c = load_cases("sampled_softmax.npz")[idx]
    q, pos_emb, pos_ids, rows, ids, table, w = _ss_inputs(c)
    kind = str(c["kind"])
    # the in-batch sampler caches NORMALISED embeddings: the table is used as is
    l2_table = bool(c["l2"]) and kind == "local"
    T = float(c["T"])
    if kind == "local":
        assert int(c["n_collisions"]) > 0 or idx == 2     # the -5e4 branch is exercised
    loss, _, _ = O.sampled_softmax_fwd(q, pos_emb, pos_ids, rows, ids, table, w, T, bool(c["l2"]), table_l2_norm=l2_table)
    np.testing.assert_allclose(loss, c["loss"], rtol=2e-5)
    dq, dpos, dtable = O.sampled_softmax_bwd(q, pos_emb, pos_ids, rows, ids, table, w, T, bool(c["l2"]), table_l2_norm=l2_table)
    np.testing.assert_allclose(dq, c["dq"], rtol=2e-4, atol=1e-5)       # fp32 reference, 513-term sums at T = 0.05
    np.testing.assert_allclose(dpos, c["dpos_emb"], rtol=2e-4, atol=2e-6)
    if kind == "local":
        np.testing.assert_allclose(dtable, c["dtable"], rtol=2e-4, atol=2e-6)


def test_oracle_matches_reference_sampled_softmax_padded_entry():
    """forward(lengths, (B, N, D) ...) == jagged_forward on the dense_to_jagged rows (sampled_softmax.py:97-193)."""
    c = load_cases("sampled_softmax.npz")[4]
    off = O.complete_cumsum(c["lengths"])
    jag = lambda x: O.dense_to_jagged(x, off)
    q, pe = jag(c["out_emb"].astype(np.float64)), jag(c["sup_emb"].astype(np.float64))
    ids = jag(c["sup_ids"][..., None].astype(np.float64))[:, 0].astype(np.int64)
    w = jag(c["sup_weights"][..., None].astype(np.float64))[:, 0]
    loss, _, _ = O.sampled_softmax_fwd(q, pe, ids, c["sampled_ids"], c["sampled_ids"], c["table"], w, float(c["T"]), True)
    np.testing.assert_allclose(loss, c["loss"], rtol=2e-5)
    dq, dpos, dtable = O.sampled_softmax_bwd(q, pe, ids, c["sampled_ids"], c["sampled_ids"], c["table"], w, float(c["T"]), True)
    np.testing.assert_allclose(dtable, c["dtable"], rtol=2e-4, atol=2e-6)
    B, N, D = c["out_emb"].shape
    np.testing.assert_allclose(O.jagged_to_padded_dense(dq, off, N), c["dout_emb"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(O.jagged_to_padded_dense(dpos, off, N), c["dsup_emb"], rtol=2e-4, atol=2e-6)


def test_dropout_generator_statistics():
    """the counter-based generator of the fused dropout, as the oracle restates it (one murmur3 finaliser per element pair): keep
    rate per tensor / row / column, neighbour / row / seed correlations -- z-scores of independent draws (tools/dropout_hash_stats.py
    prints the table)"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("dropout_hash_stats", os.path.join(os.path.dirname(__file__), "..", "tools", "dropout_hash_stats.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for seed, p in ((42, 0.1), (2 ** 40 + 7, 0.5)):
        z = mod.z_scores(seed, 1024, 1536, p)
        assert z["mean"] < 4.5 and z["column max"] < 5.5 and z["row max"] < 5.5, z
        assert all(v < 4.5 for k, v in z.items() if k not in ("mean", "column max", "row max")), z
    # an element index beyond 2^33: the pair index's high word enters the key (no repetition of the first 2^33 elements' mask)
    a = O.dropout_keep_mask(7, 1, 4096, 0.5)[0]
    assert a.shape == (1, 4096) and 0.4 < a.mean() < 0.6


def test_dropout_masks_of_two_seeds_are_not_index_permutations_of_each_other():
    """Round 5's advisor: with the seed only XORed into the pair index, mask_B[i] == mask_A[i ^ d] for EVERY pair of seeds (d = the
    XOR of their keys), and seeds with equal keys gave identical masks.  The generator now adds both seed words behind its first
    multiply: for seed pairs that differ in the low word, the high word, or both, no XOR shift of the pair index maps one mask onto
    the other -- the agreement under the old relation's d (and under every d of a scan) is what independent masks give."""
    rows, stride, p = 64, 2048, 0.5
    n_pairs = rows * stride // 2

    def pair_bits(seed):
        keep = O.dropout_keep_mask(seed, rows, stride, p)[0].reshape(-1)
        return keep[0::2].astype(np.int8) * 2 + keep[1::2].astype(np.int8)       # both uniforms of a pair index

    def old_key(seed):
        s0, s1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
        return s0 ^ (((s1 << 16) | (s1 >> 16)) & 0xFFFFFFFF)

    idx = np.arange(n_pairs)
    for a, b in ((42, 43), (42, 42 ^ (1 << 32)), (7, 7 ^ 0x10000), (2 ** 40 + 7, 2 ** 41 + 9), (5, 5 ^ (0x8000 << 32) ^ 0x8000)):
        ma, mb = pair_bits(a), pair_bits(b)
        assert not np.array_equal(ma, mb), (a, b)
        d = (old_key(a) ^ old_key(b)) & (n_pairs - 1)
        for shift in {d, 1, 2, 0x10, 0x100} | set(range(0, n_pairs, n_pairs // 64)):
            agree = float((mb == ma[idx ^ shift]).mean())
            assert agree < 0.30, (a, b, shift, agree)      # independent pairs of fair bits agree in 1/4 of the places
