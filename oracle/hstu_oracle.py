"""CPU oracle for the HSTU hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``generative_recommenders_amd/``
imports it: the product path is the HIP library and fails loudly without it.

Every function is a plain-numpy restatement (per-user loops, never visiting
padded positions) of the algorithm of one reference function; the reference
file:line it follows is cited in the docstring (paths relative to
``/root/reference/generative_recommenders``).

Pinning: the reference tree holds no golden vectors for this path
(SURVEY.md §8c).  The oracle is pinned against outputs of the reference's own
PyTorch path run in the build container -- ``tests/golden/make_golden.py``
imports ``/root/reference`` (with a 3-op fbgemm shim) and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every oracle
function against those files.  The three ``fbgemm_gpu`` ops used by the
reference (jagged_to_padded_dense / dense_to_jagged /
asynchronous_complete_cumsum; fbgemm_gpu>=1.1.0 per requirements.txt:2, not
vendored, not installed) are restated here from their published semantics and
are pinned only indirectly, through the reference call sites that consume them.
"""

from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------
# jagged / integer helpers (bit-exact domain)
# --------------------------------------------------------------------------


def complete_cumsum(lengths: np.ndarray) -> np.ndarray:
    """[0, inclusive_scan(lengths)], dtype preserved.

    Follows ops/cpp/complete_cumsum.cu:7-47 and the fbgemm op
    ``asynchronous_complete_cumsum`` used at modules/stu.py:97.
    """
    lengths = np.asarray(lengths)
    out = np.zeros(lengths.shape[0] + 1, dtype=lengths.dtype)
    np.cumsum(lengths, out=out[1:])
    return out


def jagged_to_padded_dense(
    values: np.ndarray, offsets: np.ndarray, max_len: int, padding_value: float = 0.0
) -> np.ndarray:
    """(sum L, D) -> (B, max_len, D); rows beyond max_len are dropped.

    fbgemm ``jagged_to_padded_dense`` as consumed at
    ops/pytorch/pt_hstu_attention.py:97-125.
    """
    B = offsets.shape[0] - 1
    D = values.shape[1]
    out = np.full((B, max_len, D), padding_value, dtype=values.dtype)
    for b in range(B):
        s, e = int(offsets[b]), int(offsets[b + 1])
        n = min(e - s, max_len)
        out[b, :n] = values[s : s + n]
    return out


def dense_to_jagged(dense: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """(B, N, D) -> (sum L, D), keeping the first L_b rows of each user.

    fbgemm ``dense_to_jagged`` as consumed at
    ops/pytorch/pt_hstu_attention.py:167-171.
    """
    B = offsets.shape[0] - 1
    total = int(offsets[-1])
    out = np.zeros((total, dense.shape[2]), dtype=dense.dtype)
    for b in range(B):
        s, e = int(offsets[b]), int(offsets[b + 1])
        n = min(e - s, dense.shape[1])
        out[s : s + n] = dense[b, :n]
    return out


def _dense_offsets(n_rows: int, max_len: int) -> np.ndarray:
    B = n_rows // max_len
    return (max_len * np.arange(B + 1)).astype(np.int64)


def concat_2D_jagged(
    values_left: np.ndarray,
    values_right: np.ndarray,
    max_len_left: Optional[int] = None,
    max_len_right: Optional[int] = None,
    offsets_left: Optional[np.ndarray] = None,
    offsets_right: Optional[np.ndarray] = None,
    n_prefix_from_right: int = 0,
) -> np.ndarray:
    """Per-user row concat [left_b ; right_b]; a side without offsets is dense
    with ``max_len_*`` rows per user.

    Follows ops/pytorch/pt_jagged_tensors.py:31-117 (and, for
    ``n_prefix_from_right`` > 0, pytorch_hstu_concat_l2_embeddings :208-246:
    the first ``n`` rows of the right side go in front of the left side).
    """
    if offsets_left is None:
        offsets_left = _dense_offsets(values_left.shape[0], max_len_left)
    if offsets_right is None:
        offsets_right = _dense_offsets(values_right.shape[0], max_len_right)
    B = offsets_left.shape[0] - 1
    pieces = []
    for b in range(B):
        l = values_left[int(offsets_left[b]) : int(offsets_left[b + 1])]
        r = values_right[int(offsets_right[b]) : int(offsets_right[b + 1])]
        n = n_prefix_from_right
        pieces += [r[:n], l, r[n:]]
    return np.concatenate(pieces, axis=0) if pieces else values_left[:0]


def split_2D_jagged(
    values: np.ndarray,
    max_len_left: Optional[int] = None,
    max_len_right: Optional[int] = None,
    offsets_left: Optional[np.ndarray] = None,
    offsets_right: Optional[np.ndarray] = None,
    n_prefix_to_right: int = 0,
) -> Tuple[np.ndarray, np.ndarray]:
    """Inverse of :func:`concat_2D_jagged`.

    Follows ops/pytorch/pt_jagged_tensors.py:120-205 (prefix variant:
    pytorch_hstu_split_l2_embeddings :176-205).
    """
    if offsets_left is None:
        B = offsets_right.shape[0] - 1
        offsets_left = (max_len_left * np.arange(B + 1)).astype(np.int64)
    if offsets_right is None:
        B = offsets_left.shape[0] - 1
        offsets_right = (max_len_right * np.arange(B + 1)).astype(np.int64)
    B = offsets_left.shape[0] - 1
    lefts, rights = [], []
    pos = 0
    for b in range(B):
        ll = int(offsets_left[b + 1] - offsets_left[b])
        lr = int(offsets_right[b + 1] - offsets_right[b])
        n = n_prefix_to_right
        seg = values[pos : pos + ll + lr]
        rights.append(seg[:n])
        lefts.append(seg[n : n + ll])
        rights.append(seg[n + ll :])
        pos += ll + lr
    left = np.concatenate(lefts, axis=0) if lefts else values[:0]
    right = np.concatenate(rights, axis=0) if rights else values[:0]
    return left, right


def expand_1d_jagged_to_dense(values: np.ndarray, offsets: np.ndarray, max_len: int) -> np.ndarray:
    """1-D jagged -> (B, max_len); pads with the user's LAST value (0 if empty).

    Follows ops/cpp/expand_1d_jagged_to_dense.cpp:28-52.
    """
    B = offsets.shape[0] - 1
    out = np.zeros((B, max_len), dtype=values.dtype)
    for b in range(B):
        s, e = int(offsets[b]), int(offsets[b + 1])
        n = e - s
        if n == 0:
            continue
        m = min(n, max_len)
        out[b, :m] = values[s : s + m]
        out[b, m:] = values[e - 1]
    return out


def concat_1d_jagged_jagged(
    lengths_left: np.ndarray, values_left: np.ndarray, lengths_right: np.ndarray, values_right: np.ndarray
) -> np.ndarray:
    """Per-user concat of two 1-D jagged value arrays.

    Follows ops/cpp/concat_1d_jagged_jagged.cu:33-62.
    """
    ol = complete_cumsum(np.asarray(lengths_left, dtype=np.int64))
    orr = complete_cumsum(np.asarray(lengths_right, dtype=np.int64))
    pieces = []
    for b in range(len(lengths_left)):
        pieces += [values_left[ol[b] : ol[b + 1]], values_right[orr[b] : orr[b + 1]]]
    return np.concatenate(pieces) if pieces else values_left[:0]


# --------------------------------------------------------------------------
# attention mask (integer domain)
# --------------------------------------------------------------------------


def valid_attn_mask(
    n: int,
    seq_len: int,
    num_targets: Optional[int] = None,
    max_attn_len: int = 0,
    contextual_seq_len: int = 0,
    min_full_attn_seq_len: int = 0,
) -> np.ndarray:
    """Boolean (n, n) mask of one user, rows = queries, cols = keys.

    Restates ops/pytorch/pt_hstu_attention.py:32-84 (causal branch) for a single
    user of length ``seq_len``; ``n`` may be the padded N or just ``seq_len``.
    """
    pos = np.arange(n, dtype=np.int64)
    ids = pos.copy()
    max_id = int(seq_len)
    if contextual_seq_len > 0:
        ids = np.maximum(ids - contextual_seq_len + 1, 0)
        max_id = max_id - contextual_seq_len + 1
    if num_targets is not None:
        max_id = max_id - int(num_targets)
        ids = np.minimum(ids, max_id)
    row = ids[:, None]
    col = ids[None, :]
    dist = row - col
    valid = (pos[:, None] == pos[None, :]) | (dist > 0)
    if max_attn_len > 0:
        in_window = dist <= max_attn_len
        if min_full_attn_seq_len > 0:
            in_window = in_window | (row >= max_id - min_full_attn_seq_len)
        valid = valid & in_window
    if contextual_seq_len > 0:
        valid = valid | ((row == 0) & (col < max_id))
    return valid


def _silu(x: np.ndarray) -> np.ndarray:
    return x / (1.0 + np.exp(-x))


def _sigmoid(x: np.ndarray) -> np.ndarray:
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------
# ops-path attention: forward, backward, delta-q
# --------------------------------------------------------------------------


def hstu_mha_fwd(
    max_seq_len: int,
    alpha: float,
    q: np.ndarray,
    k: np.ndarray,
    v: np.ndarray,
    seq_offsets: np.ndarray,
    num_targets: Optional[np.ndarray] = None,
    max_attn_len: int = 0,
    contextual_seq_len: int = 0,
    min_full_attn_seq_len: int = 0,
    dtype=np.float64,
) -> np.ndarray:
    """O = ((silu(alpha Q K^T) / N) * M) V per (user, head), jagged in/out.

    Restates pytorch_hstu_mha, ops/pytorch/pt_hstu_attention.py:129-171
    (padded positions contribute nothing, so they are never visited).
    q,k: (sum L, H, dqk); v: (sum L, H, dv) -> (sum L, H, dv).
    """
    q = q.astype(dtype)
    k = k.astype(dtype)
    v = v.astype(dtype)
    B = seq_offsets.shape[0] - 1
    out = np.zeros((q.shape[0], q.shape[1], v.shape[2]), dtype=dtype)
    for b in range(B):
        s, e = int(seq_offsets[b]), int(seq_offsets[b + 1])
        L = min(e - s, max_seq_len)
        if L == 0:
            continue
        t = None if num_targets is None else int(num_targets[b])
        M = valid_attn_mask(L, e - s, t, max_attn_len, contextual_seq_len, min_full_attn_seq_len)
        qb, kb, vb = q[s : s + L], k[s : s + L], v[s : s + L]
        S = alpha * np.einsum("ihd,jhd->hij", qb, kb)
        P = _silu(S) / max_seq_len * M[None]
        out[s : s + L] = np.einsum("hij,jhd->ihd", P, vb)
    return out


def hstu_mha_bwd(
    max_seq_len: int,
    alpha: float,
    dout: np.ndarray,
    q: np.ndarray,
    k: np.ndarray,
    v: np.ndarray,
    seq_offsets: np.ndarray,
    num_targets: Optional[np.ndarray] = None,
    max_attn_len: int = 0,
    contextual_seq_len: int = 0,
    min_full_attn_seq_len: int = 0,
    dtype=np.float64,
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Hand-derived backward of :func:`hstu_mha_fwd`:

        dV = P^T dO ; dP = dO V^T ; dS = dP * M / N * sig(S) (1 + S (1 - sig(S)))
        dQ = alpha dS K ; dK = alpha dS^T Q

    Cross-checks: ops/triton/triton_hstu_attention.py:968-1006,1222 and
    ops/cpp/hstu_attention/mainloop_bwd_sm80.h:886-944.
    """
    q = q.astype(dtype)
    k = k.astype(dtype)
    v = v.astype(dtype)
    dout = dout.astype(dtype)
    dq = np.zeros_like(q)
    dk = np.zeros_like(k)
    dv = np.zeros_like(v)
    B = seq_offsets.shape[0] - 1
    for b in range(B):
        s, e = int(seq_offsets[b]), int(seq_offsets[b + 1])
        L = min(e - s, max_seq_len)
        if L == 0:
            continue
        t = None if num_targets is None else int(num_targets[b])
        M = valid_attn_mask(L, e - s, t, max_attn_len, contextual_seq_len, min_full_attn_seq_len)
        qb, kb, vb, dob = q[s : s + L], k[s : s + L], v[s : s + L], dout[s : s + L]
        S = alpha * np.einsum("ihd,jhd->hij", qb, kb)
        sig = _sigmoid(S)
        P = S * sig / max_seq_len * M[None]
        dv[s : s + L] = np.einsum("hij,ihd->jhd", P, dob)
        dP = np.einsum("ihd,jhd->hij", dob, vb)
        dS = dP * M[None] / max_seq_len * sig * (1.0 + S * (1.0 - sig))
        dq[s : s + L] = alpha * np.einsum("hij,jhd->ihd", dS, kb)
        dk[s : s + L] = alpha * np.einsum("hij,ihd->jhd", dS, qb)
    return dq, dk, dv


def delta_hstu_mha_fwd(
    max_seq_len: int,
    alpha: float,
    delta_q: np.ndarray,
    k: np.ndarray,
    v: np.ndarray,
    seq_offsets: np.ndarray,
    num_targets: Optional[np.ndarray] = None,
    max_attn_len: int = 0,
    contextual_seq_len: int = 0,
    dtype=np.float64,
) -> np.ndarray:
    """Attention of the last ``delta`` rows of each user against its full K/V.

    Restates pytorch_cached_hstu_mha, ops/pytorch/pt_hstu_attention.py:174-235.
    delta_q: (B*delta, H, dqk) dense per user -> (B*delta, H, dv).
    """
    B = seq_offsets.shape[0] - 1
    delta = delta_q.shape[0] // B
    delta_q = delta_q.astype(dtype)
    k = k.astype(dtype)
    v = v.astype(dtype)
    out = np.zeros((delta_q.shape[0], delta_q.shape[1], v.shape[2]), dtype=dtype)
    for b in range(B):
        s, e = int(seq_offsets[b]), int(seq_offsets[b + 1])
        L = e - s
        t = None if num_targets is None else int(num_targets[b])
        M = valid_attn_mask(L, L, t, max_attn_len, contextual_seq_len, 0)[L - delta :]
        qb = delta_q[b * delta : (b + 1) * delta]
        kb, vb = k[s:e], v[s:e]
        S = alpha * np.einsum("ihd,jhd->hij", qb, kb)
        P = _silu(S) / max_seq_len * M[None]
        out[b * delta : (b + 1) * delta] = np.einsum("hij,jhd->ihd", P, vb)
    return out


# --------------------------------------------------------------------------
# research-path attention with relative position / bucketed-time bias
# --------------------------------------------------------------------------


def rel_time_buckets(ts_row: np.ndarray, n: int) -> np.ndarray:
    """bucket[i, j] = clamp(floor(log(max(|ext[i+1] - ext[j]|, 1)) / 0.301), 0, 128)
    with ext = concat(ts[0..n-1], ts[n-1]).

    Restates research/modeling/sequential/hstu.py:128-139 (bucketization fn
    :610-612).  Returns int64 (n, n).
    """
    ts_row = np.asarray(ts_row[:n], dtype=np.int64)
    ext = np.concatenate([ts_row, ts_row[n - 1 : n]])
    diff = ext[1:, None] - ext[None, :-1]
    val = np.log(np.maximum(np.abs(diff), 1).astype(np.float32)) / np.float32(0.301)
    return np.clip(val.astype(np.int64), 0, 128)


def rel_bias_attention_fwd(
    n: int,
    q: np.ndarray,
    k: np.ndarray,
    v: np.ndarray,
    seq_offsets: np.ndarray,
    timestamps: Optional[np.ndarray],
    pos_w: np.ndarray,
    ts_w: Optional[np.ndarray],
    dtype=np.float64,
) -> np.ndarray:
    """Research-path attention: P = silu(q.k + pos_w[n-1+j-i] + ts_w[bkt]) / n * tril.

    Restates _hstu_attention_maybe_from_cache + RelativeBucketedTimeAndPositionBasedBias,
    research/modeling/sequential/hstu.py:150-223, 87-144.
    q,k: (sum L, H, dqk), v: (sum L, H, dv), timestamps: (B, n) int64.
    """
    q = q.astype(dtype)
    k = k.astype(dtype)
    v = v.astype(dtype)
    B = seq_offsets.shape[0] - 1
    out = np.zeros((q.shape[0], q.shape[1], v.shape[2]), dtype=dtype)
    for b in range(B):
        s, e = int(seq_offsets[b]), int(seq_offsets[b + 1])
        L = min(e - s, n)
        if L == 0:
            continue
        i = np.arange(L)
        bias = pos_w.astype(dtype)[(n - 1) + i[None, :] - i[:, None]]
        if ts_w is not None:
            bias = bias + ts_w.astype(dtype)[rel_time_buckets(timestamps[b], n)[:L, :L]]
        S = np.einsum("ihd,jhd->hij", q[s : s + L], k[s : s + L]) + bias[None]
        P = _silu(S) / n * (i[:, None] >= i[None, :])[None]
        out[s : s + L] = np.einsum("hij,jhd->ihd", P, v[s : s + L])
    return out


def rel_bias_attention_bwd(
    n: int,
    dout: np.ndarray,
    q: np.ndarray,
    k: np.ndarray,
    v: np.ndarray,
    seq_offsets: np.ndarray,
    timestamps: Optional[np.ndarray],
    pos_w: np.ndarray,
    ts_w: Optional[np.ndarray],
    dtype=np.float64,
):
    """Backward of :func:`rel_bias_attention_fwd`; also returns d pos_w, d ts_w
    (scatter-adds of dS over heads and users)."""
    q = q.astype(dtype)
    k = k.astype(dtype)
    v = v.astype(dtype)
    dout = dout.astype(dtype)
    dq, dk, dv = np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)
    dpos = np.zeros(pos_w.shape, dtype=dtype)
    dts = None if ts_w is None else np.zeros(ts_w.shape, dtype=dtype)
    B = seq_offsets.shape[0] - 1
    for b in range(B):
        s, e = int(seq_offsets[b]), int(seq_offsets[b + 1])
        L = min(e - s, n)
        if L == 0:
            continue
        i = np.arange(L)
        pidx = (n - 1) + i[None, :] - i[:, None]
        bias = pos_w.astype(dtype)[pidx]
        if ts_w is not None:
            bkt = rel_time_buckets(timestamps[b], n)[:L, :L]
            bias = bias + ts_w.astype(dtype)[bkt]
        tril = (i[:, None] >= i[None, :])[None]
        S = np.einsum("ihd,jhd->hij", q[s : s + L], k[s : s + L]) + bias[None]
        sig = _sigmoid(S)
        P = S * sig / n * tril
        dv[s : s + L] = np.einsum("hij,ihd->jhd", P, dout[s : s + L])
        dP = np.einsum("ihd,jhd->hij", dout[s : s + L], v[s : s + L])
        dS = dP * tril / n * sig * (1.0 + S * (1.0 - sig))
        dq[s : s + L] = np.einsum("hij,jhd->ihd", dS, k[s : s + L])
        dk[s : s + L] = np.einsum("hij,ihd->jhd", dS, q[s : s + L])
        dSh = dS.sum(axis=0)
        np.add.at(dpos, pidx, dSh)
        if ts_w is not None:
            np.add.at(dts, bkt, dSh)
    return dq, dk, dv, dpos, dts


# --------------------------------------------------------------------------
# norms and the projections around attention
# --------------------------------------------------------------------------


def layer_norm_fwd(x, weight, bias, eps, dtype=np.float64):
    """Row LayerNorm with affine, math in ``dtype``.  ops/pytorch/pt_layer_norm.py:24-38."""
    x = x.astype(dtype)
    mean = x.mean(axis=1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    return (x - mean) * rstd * weight.astype(dtype) + bias.astype(dtype)


def layer_norm_bwd(dy, x, weight, eps, dtype=np.float64):
    """dx, dweight, dbias of :func:`layer_norm_fwd`."""
    x = x.astype(dtype)
    dy = dy.astype(dtype)
    w = weight.astype(dtype)
    mean = x.mean(axis=1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * rstd
    g = dy * w
    dx = rstd * (g - g.mean(axis=1, keepdims=True) - xhat * (g * xhat).mean(axis=1, keepdims=True))
    return dx, (dy * xhat).sum(axis=0), dy.sum(axis=0)


def swish_layer_norm_fwd(x, weight, bias, eps, dtype=np.float64):
    """y = x * sigmoid(LayerNorm(x)), math in ``dtype``.  ops/pytorch/pt_layer_norm.py:41-62 (the gate in front of the
    preprocessors' and DlrmHSTU's MLPs: modules/preprocessors.py:160,181, modules/dlrm_hstu.py:144,240)."""
    x = x.astype(dtype)
    z = layer_norm_fwd(x, weight, bias, eps, dtype)
    return x / (1.0 + np.exp(-z))


def swish_layer_norm_bwd(dy, x, weight, bias, eps, dtype=np.float64):
    """dx, dweight, dbias of :func:`swish_layer_norm_fwd`: with s = sigmoid(z), d z = dy x s (1 - s) goes through the
    LayerNorm's backward, and dy s reaches x directly."""
    x = x.astype(dtype)
    dy = dy.astype(dtype)
    s = 1.0 / (1.0 + np.exp(-layer_norm_fwd(x, weight, bias, eps, dtype)))
    dz = dy * x * s * (1.0 - s)
    dx, dw, db = layer_norm_bwd(dz, x, weight, eps, dtype)
    return dx + dy * s, dw, db


def group_norm_rows(x, weight, bias, eps, num_heads, linear_dim, dtype=np.float64):
    """Per-row, per-head normalisation with one (weight, bias) scalar per head.

    F.group_norm on (-1, H, linear_dim) with num_groups=H as used at
    ops/pytorch/pt_hstu_linear.py:43-50."""
    x = x.astype(dtype).reshape(-1, num_heads, linear_dim)
    mean = x.mean(axis=2, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=2, keepdims=True)
    y = (x - mean) / np.sqrt(var + eps)
    y = y * weight.astype(dtype)[None, :, None] + bias.astype(dtype)[None, :, None]
    return y.reshape(-1, num_heads * linear_dim)


def hstu_compute_uqvk(x, norm_weight, norm_bias, norm_eps, num_heads, attn_dim, hidden_dim,
                      uvqk_weight, uvqk_bias, dtype=np.float64):
    """LN_affine(x) @ W + b, split [u, v, q, k], silu on u only.

    Restates ops/hstu_compute.py:50-89."""
    nx = layer_norm_fwd(x, norm_weight, norm_bias, norm_eps, dtype)
    uvqk = nx @ uvqk_weight.astype(dtype) + uvqk_bias.astype(dtype)
    hv, ha = hidden_dim * num_heads, attn_dim * num_heads
    u = _silu(uvqk[:, :hv])
    v = uvqk[:, hv : 2 * hv].reshape(-1, num_heads, hidden_dim)
    q = uvqk[:, 2 * hv : 2 * hv + ha].reshape(-1, num_heads, attn_dim)
    k = uvqk[:, 2 * hv + ha :].reshape(-1, num_heads, attn_dim)
    return u, q, k, v


def norm_mul(attn, u, weight, bias, eps, concat_ux, group_norm, num_heads, linear_dim, dtype=np.float64):
    """y = u * LN_or_GN(attn), optionally concat [u, attn, y] (dropout off).

    Restates pytorch_norm_mul_dropout, ops/pytorch/pt_hstu_linear.py:23-65."""
    attn = attn.astype(dtype)
    u = u.astype(dtype)
    if group_norm:
        n = group_norm_rows(attn, weight, bias, eps, num_heads, linear_dim, dtype)
    else:
        n = layer_norm_fwd(attn, weight, bias, eps, dtype)
    y = u * n
    if concat_ux:
        y = np.concatenate([u, attn, y], axis=1)
    return y


def hstu_compute_output(attn, u, x, norm_weight, norm_bias, norm_eps, output_weight,
                        num_heads, linear_dim, concat_ux, group_norm, dtype=np.float64):
    """x + norm_mul(attn, u) @ W_o.  Restates ops/pytorch/pt_hstu_linear.py:68-99."""
    y = norm_mul(attn, u, norm_weight, norm_bias, norm_eps, concat_ux, group_norm, num_heads, linear_dim, dtype)
    return x.astype(dtype) + y @ output_weight.astype(dtype)


def dropout_keep_mask(seed: int, rows: int, out_stride: int, dropout_ratio: float) -> Tuple[np.ndarray, float]:
    """Keep mask (bool, (rows, out_stride)) and survivor scale of the fused output-stage dropout.

    What is restated: the SEMANTICS of the reference's in-kernel dropout (triton_hstu_linear.py:101-120 /
    pt_hstu_linear.py:60-64: every element of the concatenated output is dropped independently with probability p and
    survivors are scaled by 1 / (1 - p); the backward reuses the mask) and, bit for bit, the counter-based generator of
    the HIP kernels (csrc/norm_kernels.inc, drop_hash) -- the reference's own Philox stream (tl.rand / F.dropout) is not
    reproducible across implementations, so parity under dropout is (a) this exact mask and (b) the statistics.
    Element e = row * out_stride + col; pair j = e >> 1 shares one 32-bit hash (murmur3's fmix32 of the pair index xor a key, the seed's high word added behind the first multiply), element e takes its low (e even) or
    high (e odd) 16 bits; keep iff r16 >= thr, thr = clamp(round(p * 65536), 1, 65535); scale = 65536 / (65536 - thr).
    """
    thr = int(min(max(np.floor(float(np.float32(dropout_ratio)) * 65536.0 + 0.5), 1.0), 65535.0)) if dropout_ratio > 0 else 0
    n = rows * out_stride
    e = np.arange(n, dtype=np.uint64)
    pe = e >> np.uint64(1)
    lo = (pe & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    hi = (pe >> np.uint64(32)).astype(np.uint64)
    s0, s1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    M = np.uint64(0xFFFFFFFF)

    def mul(a, c):
        return (a * np.uint64(c)) & M

    k = (s1 + mul(hi, 0x9E3779B9)) & M
    key = s0 ^ (((k << np.uint64(16)) | (k >> np.uint64(16))) & M)          # both seed words and the index's high word
    h = lo ^ key
    # (both seed words once more behind the first multiply: without it the masks of two seeds are index-XOR permutations of each other)
    h ^= h >> np.uint64(16); h = mul(h, 0x85EBCA6B); h = (h + (s1 ^ (((s0 << np.uint64(13)) | (s0 >> np.uint64(19))) & M))) & M; h ^= h >> np.uint64(13); h = mul(h, 0xC2B2AE35); h ^= h >> np.uint64(16)
    r16 = np.where((e & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    keep = (r16 >= np.uint64(thr)).reshape(rows, out_stride)
    return keep, 65536.0 / (65536.0 - thr)


# --------------------------------------------------------------------------
# synthetic length generators (measurement inputs)
# --------------------------------------------------------------------------


def generate_sparse_seq_len(rng: np.random.Generator, size: int, max_seq_len: int, sparsity: float) -> np.ndarray:
    """Length distribution of common.py:173-201 (numpy RNG, so streams differ)."""
    if sparsity == 0.0:
        return np.zeros(size, dtype=np.int32)
    if sparsity == 1.0:
        return np.full(size, max_seq_len, dtype=np.int32)
    if sparsity >= 0.5:
        lo = int((2 * sparsity - 1.0) * max_seq_len)
        return rng.integers(lo, max_seq_len, size=size, dtype=np.int32)
    hi = int(2 * sparsity * max_seq_len)
    return rng.integers(0, hi, size=size, dtype=np.int32)


# ---------------------------------------------------------------------------
# timestamp / position additive encoder
# reference: ops/position.py:38-96, ops/pytorch/pt_position.py:40-134, caller modules/positional_encoder.py:52-75
# ---------------------------------------------------------------------------
def position_indices(length, num_target, max_contextual_seq_len, max_pos_ind, interleave_targets):
    """position-table row of every row r < length of one user (pt_position.py:40-73).  `num_target` None = no
    targets (then index = length - r: the reference's PyTorch branch only survives full-length batches there,
    because it also indexes the table for its padded columns)."""
    r = np.arange(length, dtype=np.int64)
    if num_target is not None:
        high = length - num_target * (2 if interleave_targets else 1)
        idx = high - np.minimum(r, high)
    else:
        idx = length - r
    idx = np.minimum(idx + max_contextual_seq_len, max_pos_ind - 1)
    if max_contextual_seq_len > 0:
        c = min(max_contextual_seq_len, length)
        idx[:c] = np.arange(c)
    return idx


def time_bucket_indices(ts_user, max_bucket, time_bucket_fn):
    """bucket of (query time - timestamp) per row of one user (pt_position.py:100-122): fp32 arithmetic exactly as
    torch does it -- int64 difference -> fp32, clamp(min=1e-6), / 60, sqrt | log, truncate, clamp to
    [0, max_bucket].  The caller chooses max_bucket: the reference's PyTorch path takes ts_embeddings.size(1) - 1 (the
    embedding DIM minus one, `:101`), its GPU path the last table row (triton_position.py:275,295)."""
    if len(ts_user) == 0:
        return np.zeros(0, dtype=np.int64)
    d = (ts_user[-1] - ts_user).astype(np.float32)
    t = np.maximum(d, np.float32(1e-6)) / np.float32(60.0)
    t = np.log(t) if time_bucket_fn == "log" else np.sqrt(t)
    t = np.maximum(t / np.float32(1.0), np.float32(0.0))
    return np.clip(t.astype(np.int32).astype(np.int64), 0, max_bucket)


def add_timestamp_positional_embeddings_fwd(alpha, x, seq_offsets, timestamps, pos_w, ts_w, max_contextual_seq_len,
                                            num_targets, interleave_targets, time_bucket_fn, bucket_clamp="table"):
    """out[row] = alpha * x[row] + pos_w[pos_idx] + ts_w[ts_idx]  (ops/position.py:55, pt_position.py:123-134).
    Returns (out, pos_idx, ts_idx) with the per-row table indices.  bucket_clamp: "table" = the GPU path's clamp to the
    last table row (triton_position.py:275,295), "pytorch_path" = min(D - 1, last row) (pt_position.py:101; an index
    past the table would raise there)."""
    max_bucket = ts_w.shape[0] - 1 if bucket_clamp == "table" else min(ts_w.shape[1] - 1, ts_w.shape[0] - 1)
    B = len(seq_offsets) - 1
    pos_idx = np.zeros(x.shape[0], dtype=np.int64)
    ts_idx = np.zeros(x.shape[0], dtype=np.int64)
    for b in range(B):
        o, e = int(seq_offsets[b]), int(seq_offsets[b + 1])
        nt = None if num_targets is None else int(num_targets[b])
        pos_idx[o:e] = position_indices(e - o, nt, max_contextual_seq_len, pos_w.shape[0], interleave_targets)
        ts_idx[o:e] = time_bucket_indices(np.asarray(timestamps[o:e], dtype=np.int64), max_bucket, time_bucket_fn)
    out = x * alpha + (ts_w[ts_idx] + pos_w[pos_idx])
    return out, pos_idx, ts_idx


def add_timestamp_positional_embeddings_bwd(alpha, g, pos_idx, ts_idx, n_pos, n_ts):
    """d x = alpha g;  d pos_w[i] = sum of g rows with pos_idx == i;  same for ts_w (index_select backward)."""
    dpos = np.zeros((n_pos, g.shape[1]), dtype=np.float64)
    dts = np.zeros((n_ts, g.shape[1]), dtype=np.float64)
    np.add.at(dpos, pos_idx, g)
    np.add.at(dts, ts_idx, g)
    return g * alpha, dpos, dts


# ---------------------------------------------------------------------------
# output post-processing
# reference: modules/postprocessors.py:55-69 (l2 norm), modules/hstu_transducer.py:191-251 (candidate split)
# ---------------------------------------------------------------------------
def l2_norm_fwd(x, eps=1e-6):
    """x / max(||x||_2, eps) per row (postprocessors.py:66-68)."""
    n = np.sqrt((x * x).sum(axis=-1, keepdims=True))
    return x / np.maximum(n, eps)


def l2_norm_bwd(g, x, eps=1e-6):
    """n > eps: (g - y <y, g>) / n;  n <= eps (clamp active, zero gradient through it): g / eps."""
    n = np.sqrt((x * x).sum(axis=-1, keepdims=True))
    y = x / np.maximum(n, eps)
    return np.where(n > eps, (g - y * (y * g).sum(axis=-1, keepdims=True)) / np.maximum(n, eps), g / eps)


def split_candidates(values, lengths, num_targets):
    """rows of the last num_targets[b] positions of every user, concatenated (hstu_transducer.py:207-221: the right
    part of split_2D_jagged with uih lengths = lengths - num_targets).  Returns (candidates, source row indices)."""
    off = complete_cumsum(np.asarray(lengths, dtype=np.int64))
    idx = np.concatenate([np.arange(off[b + 1] - int(num_targets[b]), off[b + 1]) for b in range(len(lengths))]).astype(np.int64)
    return values[idx], idx


# ---------------------------------------------------------------------------
# sampled-softmax loss with a dot-product similarity
# reference: research/modeling/sequential/losses/sampled_softmax.py:44-95 (jagged_forward),
#            autoregressive_losses.py:37-45 (_maybe_l2_norm), :112-131 (LocalNegativesSampler.forward),
#            rails/similarities/dot_product_similarity_fn.py:38-62
# ---------------------------------------------------------------------------

def _l2_normalize_rows(x, eps):
    """x / clamp(||x||_2, min=eps) per row (autoregressive_losses.py:40-44)."""
    nrm = np.sqrt((x * x).sum(axis=-1, keepdims=True))
    return x / np.maximum(nrm, eps), nrm


def _l2_normalize_rows_bwd(dy, x, eps):
    """Gradient of _l2_normalize_rows: the clamp passes no gradient to the norm where ||x|| < eps."""
    y, nrm = _l2_normalize_rows(x, eps)
    c = np.maximum(nrm, eps)
    proj = (y * dy).sum(axis=-1, keepdims=True)
    return np.where(nrm >= eps, (dy - y * proj) / c, dy / c)


def sampled_softmax_fwd(q, pos_emb, pos_ids, neg_rows, neg_ids, table, weights, temperature, l2_norm, eps=1e-6,
                        dtype=np.float64, table_l2_norm=None):
    """loss = sum_i w_i * (-log_softmax([l_i0, l_i1..l_iR])[0]) / sum_i w_i with
    l_i0 = <q_i, norm(pos_emb_i)> / T,  l_ik = <q_i, norm(table[neg_rows_ik])> / T, and l_ik = -5e4 where
    neg_ids_ik == pos_ids_i (sampled_softmax.py:60-93).  ``neg_rows`` index ``table``; ``neg_ids`` are the item ids
    of those rows (identical for the local sampler, cached ids for the in-batch one).
    Returns (loss, per-row loss, per-row logsumexp)."""
    q, pos_emb, table = q.astype(dtype), pos_emb.astype(dtype), table.astype(dtype)
    n, R = neg_rows.shape
    tl2 = l2_norm if table_l2_norm is None else table_l2_norm    # in-batch sampler: the table is already normalised
    pos_hat = _l2_normalize_rows(pos_emb, eps)[0] if l2_norm else pos_emb
    row_loss = np.zeros(n, dtype=dtype)
    lse = np.zeros(n, dtype=dtype)
    for i in range(n):
        neg = table[neg_rows[i]]
        neg_hat = _l2_normalize_rows(neg, eps)[0] if tl2 else neg
        logits = np.empty(R + 1, dtype=dtype)
        logits[0] = (q[i] * pos_hat[i]).sum() / temperature
        logits[1:] = np.where(neg_ids[i] == pos_ids[i], -5e4, neg_hat @ q[i] / temperature)
        m = logits.max()
        lse[i] = m + np.log(np.exp(logits - m).sum())
        row_loss[i] = lse[i] - logits[0]
    w = weights.astype(dtype)
    return (row_loss * w).sum() / w.sum(), row_loss, lse


def sampled_softmax_bwd(q, pos_emb, pos_ids, neg_rows, neg_ids, table, weights, temperature, l2_norm, eps=1e-6,
                        dtype=np.float64, table_l2_norm=None):
    """Hand-derived gradients of sampled_softmax_fwd's scalar loss w.r.t. q, pos_emb and table
    (d loss / d l_i0 = g_i (p_i0 - 1), d loss / d l_ik = g_i p_ik for unmasked k, g_i = w_i / sum w)."""
    q, pos_emb, table = q.astype(dtype), pos_emb.astype(dtype), table.astype(dtype)
    n, R = neg_rows.shape
    tl2 = l2_norm if table_l2_norm is None else table_l2_norm
    w = weights.astype(dtype)
    g = w / w.sum()
    pos_hat = _l2_normalize_rows(pos_emb, eps)[0] if l2_norm else pos_emb
    dq = np.zeros_like(q)
    dpos_hat = np.zeros_like(pos_emb)
    dtable = np.zeros_like(table)
    for i in range(n):
        neg = table[neg_rows[i]]
        neg_hat = _l2_normalize_rows(neg, eps)[0] if tl2 else neg
        masked = neg_ids[i] == pos_ids[i]
        logits = np.empty(R + 1, dtype=dtype)
        logits[0] = (q[i] * pos_hat[i]).sum() / temperature
        logits[1:] = np.where(masked, -5e4, neg_hat @ q[i] / temperature)
        m = logits.max()
        p = np.exp(logits - m)
        p /= p.sum()
        dl = g[i] * p
        dl[0] -= g[i]
        dl[1:][masked] = 0.0                       # torch.where picks the constant: no gradient to the similarity
        dq[i] = (dl[0] * pos_hat[i] + dl[1:] @ neg_hat) / temperature
        dpos_hat[i] = dl[0] * q[i] / temperature
        dneg_hat = dl[1:, None] * q[i][None, :] / temperature
        dneg = _l2_normalize_rows_bwd(dneg_hat, neg, eps) if tl2 else dneg_hat
        np.add.at(dtable, neg_rows[i], dneg)
    dpos = _l2_normalize_rows_bwd(dpos_hat, pos_emb, eps) if l2_norm else dpos_hat
    return dq, dpos, dtable
