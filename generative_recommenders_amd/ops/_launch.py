"""Tensor -> C-ABI marshalling for libhstu_hip.so.  No math here: every function packs
pointers / strides / sizes, launches on torch's current stream, and returns torch tensors
it allocated for the outputs."""

from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from generative_recommenders_amd import _lib as L


# Outputs are torch.empty and the kernels write what the jagged metadata describes -- nothing else.  Rows the metadata
# does NOT describe stay uninitialised (the reference's padded-dense PyTorch path would truncate / zero them): attention
# output rows past seq_offsets[-1], delta-attention rows of a user shorter than delta_q, concat / split rows of a user
# whose left + right length exceeds max_seq_len.  All three are caller errors; HSTU_DEBUG_CHECKS=1 turns them into
# exceptions (one host sync per call, debugging only).
DEBUG_CHECKS = os.environ.get("HSTU_DEBUG_CHECKS", "0") not in ("", "0")


def _check_attention_metadata(q, seq_offsets, max_seq_len, delta_q) -> None:
    lengths = (seq_offsets[1:] - seq_offsets[:-1])
    if bool((lengths < 0).any()):
        raise RuntimeError("seq_offsets must be non-decreasing")
    if delta_q > 0:
        if bool((lengths < delta_q).any()):
            raise RuntimeError(f"delta attention: a user has fewer than delta_q = {delta_q} rows (its output rows would stay uninitialised)")
    elif int(seq_offsets[-1]) != q.shape[0]:
        raise RuntimeError(f"seq_offsets[-1] = {int(seq_offsets[-1])} but q has {q.shape[0]} rows (the rows beyond would stay uninitialised)")
    if bool((lengths > max_seq_len).any()):
        raise RuntimeError(f"a user is longer than max_seq_len = {max_seq_len}")


def _check_pair_lengths(ol, orr, max_len_left, max_len_right, max_seq_len, batch) -> None:
    def lens(o, m):
        return (o[1:] - o[:-1]) if o is not None else torch.full((batch,), int(m), dtype=torch.int64, device=(ol if ol is not None else orr).device)
    tot = lens(ol, max_len_left).to(torch.int64) + lens(orr, max_len_right).to(torch.int64)
    if bool((tot > max_seq_len).any()):
        raise RuntimeError(f"concat / split: a user's left + right length exceeds max_seq_len = {max_seq_len} (rows beyond it are not written)")


def _vp(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _aligned_rows(t: torch.Tensor) -> torch.Tensor:
    """(rows, H, d) tensor whose last dim is contiguous and whose (row, head) vectors start
    16-byte aligned; copies only when the layout forces it (flash_common.cpp:360-365)."""
    es = t.element_size()
    ok = (
        t.stride(-1) == 1
        and (t.stride(0) * es) % 16 == 0
        and (t.stride(1) * es) % 16 == 0
        and t.data_ptr() % 16 == 0
    )
    return t if ok else t.contiguous()


def _idx(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype not in (torch.int32, torch.int64):
        t = t.to(torch.int64)
    return t.contiguous()


_ORDER_CACHE = {}


def length_order(seq_offsets: torch.Tensor) -> torch.Tensor:
    """users by descending length (int32 permutation) -- the launch order of the reference's ``sort_by_length``
    (ops/triton/triton_hstu_attention.py:1968-1973) -- at the granularity the kernels' work has: the number of 32-row
    tiles, ties in batch order.  (Sorted by the exact length, a batch of near-equal lengths -- 180..199 rows: 6 or 7 tiles
    -- is walked in a random order for no gain in balance: measured +5 % forward / +3 % backward at 1024 users against
    batch order; by tiles it is two runs in batch order.)  One argsort per batch: the layers of a stack call with the same
    offsets TENSOR OBJECT, so the last result is kept, keyed on that object (a weak reference: alive and the same
    version).  Keying on the storage address would hand the next batch -- whose freshly built offsets the caching
    allocator likes to put at the same address -- the previous batch's order: still a valid permutation, silently the
    wrong one."""
    import weakref

    hit = _ORDER_CACHE.get("last")
    if hit is not None and hit[0]() is seq_offsets and hit[1] == seq_offsets._version:
        return hit[2]
    tiles = (seq_offsets[1:] - seq_offsets[:-1] + 31) >> 5
    order = torch.argsort(tiles, descending=True, stable=True).to(torch.int32)
    _ORDER_CACHE["last"] = (weakref.ref(seq_offsets), seq_offsets._version, order)
    return order


def _fill_attn_params(p: L.HstuAttnParams, q, k, v, out, seq_offsets, num_targets, max_seq_len, alpha, scale,
                      max_attn_len, contextual_seq_len, min_full_attn_seq_len, delta_q, user_order=None) -> None:
    p.user_order = _vp(user_order)
    p.q, p.k, p.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    p.out = _vp(out)
    p.seq_offsets = seq_offsets.data_ptr()
    p.num_targets = _vp(num_targets)
    p.q_row_stride, p.q_head_stride = q.stride(0), q.stride(1)
    p.k_row_stride, p.k_head_stride = k.stride(0), k.stride(1)
    p.v_row_stride, p.v_head_stride = v.stride(0), v.stride(1)
    if out is not None:
        p.o_row_stride, p.o_head_stride = out.stride(0), out.stride(1)
    p.batch = seq_offsets.numel() - 1
    p.heads = q.shape[1]
    p.dqk, p.dv = q.shape[2], v.shape[2]
    p.max_seq_len = int(max_seq_len)
    p.delta_q = int(delta_q)
    p.alpha = float(alpha)
    p.scale = float(scale)
    p.max_attn_len = int(max_attn_len)
    p.contextual_seq_len = int(contextual_seq_len)
    p.min_full_attn_seq_len = int(min_full_attn_seq_len)
    p.dtype = L.torch_dtype_code(q.dtype)
    p.offsets_dtype = L.index_dtype_code(seq_offsets)
    p.targets_dtype = L.index_dtype_code(num_targets) if num_targets is not None else 0


def attn_fwd(q, k, v, seq_offsets, num_targets, max_seq_len, alpha, scale, max_attn_len=0,
             contextual_seq_len=0, min_full_attn_seq_len=0, delta_q=0, user_order=None) -> torch.Tensor:
    for name, t in (("q", q), ("k", k), ("v", v), ("seq_offsets", seq_offsets)):
        L.require_gpu_tensor(t, name)
    if not (q.dtype == k.dtype == v.dtype):
        raise RuntimeError("q, k, v must have the same dtype")
    q, k, v = _aligned_rows(q), _aligned_rows(k), _aligned_rows(v)
    seq_offsets, num_targets = _idx(seq_offsets), _idx(num_targets)
    out = torch.empty((q.shape[0], q.shape[1], v.shape[2]), dtype=q.dtype, device=q.device)
    if q.shape[0] == 0:
        return out
    if DEBUG_CHECKS:
        _check_attention_metadata(q, seq_offsets, max_seq_len, delta_q)
    p = L.HstuAttnParams()
    _fill_attn_params(p, q, k, v, out, seq_offsets, num_targets, max_seq_len, alpha, scale, max_attn_len,
                      contextual_seq_len, min_full_attn_seq_len, delta_q, user_order)
    with torch.cuda.device(q.device):
        L.check(L.lib().hstu_attn_fwd(C.byref(p), L.current_stream_ptr(q.device)))
    return out


def attn_bwd(dout, q, k, v, seq_offsets, num_targets, max_seq_len, alpha, scale, max_attn_len=0,
             contextual_seq_len=0, min_full_attn_seq_len=0,
             dq: Optional[torch.Tensor] = None, dk: Optional[torch.Tensor] = None,
             dv: Optional[torch.Tensor] = None, user_order=None,
             deterministic: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """dq/dk/dv may be pre-allocated (possibly strided views of one fused buffer), as in
    hstu::hstu_mha_bwd (flash_api.cpp:111-141).  ``deterministic`` (default: torch's own switch,
    ``torch.are_deterministic_algorithms_enabled()``): sequences that need several key blocks add their dq partials in block
    order instead of with atomics (HstuAttnBwdParams::deterministic)."""
    for name, t in (("dout", dout), ("q", q), ("k", k), ("v", v)):
        L.require_gpu_tensor(t, name)
    q, k, v, dout = _aligned_rows(q), _aligned_rows(k), _aligned_rows(v), _aligned_rows(dout)
    seq_offsets, num_targets = _idx(seq_offsets), _idx(num_targets)
    dq = torch.empty_like(q, memory_format=torch.contiguous_format) if dq is None else dq
    dk = torch.empty_like(k, memory_format=torch.contiguous_format) if dk is None else dk
    dv = torch.empty_like(v, memory_format=torch.contiguous_format) if dv is None else dv
    if q.shape[0] == 0:
        return dq, dk, dv
    bp = L.HstuAttnBwdParams()
    _fill_attn_params(bp.fwd, q, k, v, None, seq_offsets, num_targets, max_seq_len, alpha, scale, max_attn_len,
                      contextual_seq_len, min_full_attn_seq_len, 0, user_order)
    bp.dout, bp.dq, bp.dk, bp.dv = dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    bp.do_row_stride, bp.do_head_stride = dout.stride(0), dout.stride(1)
    bp.dq_row_stride, bp.dq_head_stride = dq.stride(0), dq.stride(1)
    bp.dk_row_stride, bp.dk_head_stride = dk.stride(0), dk.stride(1)
    bp.dv_row_stride, bp.dv_head_stride = dv.stride(0), dv.stride(1)
    bp.total_rows = q.shape[0]
    bp.deterministic = int(torch.are_deterministic_algorithms_enabled() if deterministic is None else deterministic)
    ws_bytes = L.lib().hstu_attn_bwd_workspace_bytes(C.byref(bp))
    ws = None
    if ws_bytes:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
        bp.workspace = ws.data_ptr()
    with torch.cuda.device(q.device):
        L.check(L.lib().hstu_attn_bwd(C.byref(bp), L.current_stream_ptr(q.device)))
    return dq, dk, dv


def _shape_params(p: L.HstuAttnParams, dtype, heads, dqk, dv, max_seq_len, alpha, max_attn_len, contextual_seq_len,
                  min_full_attn_seq_len, with_bias) -> None:
    p.batch, p.heads, p.dqk, p.dv, p.max_seq_len = 1, int(heads), int(dqk), int(dv), int(max_seq_len)
    p.alpha, p.scale = float(alpha), 1.0 / max_seq_len
    p.max_attn_len, p.contextual_seq_len, p.min_full_attn_seq_len = int(max_attn_len), int(contextual_seq_len), int(min_full_attn_seq_len)
    p.dtype = L.torch_dtype_code(dtype)
    p.pos_w = 1 if with_bias else None          # only NULL / non-NULL is looked at
    if with_bias and with_bias != "position":   # True: position AND time tables (128 buckets unless with_bias is a number)
        p.ts_w, p.timestamps = 1, 1
        p.num_buckets = 128 if with_bias is True else int(with_bias)


def attn_fwd_kernel_name(dtype, dqk, dv, max_seq_len, heads=1, alpha=1.0, max_attn_len=0, contextual_seq_len=0,
                         min_full_attn_seq_len=0, with_bias=False) -> str:
    """the forward instantiation the library dispatches for this shape (hstu_attn_fwd_kernel_name)"""
    p = L.HstuAttnParams()
    _shape_params(p, dtype, heads, dqk, dv, max_seq_len, alpha, max_attn_len, contextual_seq_len, min_full_attn_seq_len, with_bias)
    buf = C.create_string_buffer(128)
    L.check(L.lib().hstu_attn_fwd_kernel_name(C.byref(p), buf, 128))
    return buf.value.decode()


def attn_bwd_kernel_name(dtype, dqk, dv, max_seq_len, heads=1, alpha=1.0, max_attn_len=0, contextual_seq_len=0,
                         min_full_attn_seq_len=0, with_bias=False) -> str:
    """the backward instantiation the library dispatches for this shape (hstu_attn_bwd_kernel_name)"""
    bp = L.HstuAttnBwdParams()
    _shape_params(bp.fwd, dtype, heads, dqk, dv, max_seq_len, alpha, max_attn_len, contextual_seq_len, min_full_attn_seq_len,
                  with_bias)
    buf = C.create_string_buffer(128)
    L.check(L.lib().hstu_attn_bwd_kernel_name(C.byref(bp), buf, 128))
    return buf.value.decode()


# ----------------------------------------------------------------------------- jagged
def complete_cumsum(lengths: torch.Tensor) -> torch.Tensor:
    L.require_gpu_tensor(lengths, "lengths")
    lengths = _idx(lengths)
    out = torch.empty(lengths.numel() + 1, dtype=lengths.dtype, device=lengths.device)
    with torch.cuda.device(lengths.device):
        L.check(L.lib().hstu_complete_cumsum(lengths.data_ptr(), out.data_ptr(), lengths.numel(),
                                             L.index_dtype_code(lengths), L.current_stream_ptr(lengths.device)))
    return out


def _pair_offsets(offsets_left, offsets_right):
    ol, orr = _idx(offsets_left), _idx(offsets_right)
    if ol is not None and orr is not None and ol.dtype != orr.dtype:
        ol, orr = ol.to(torch.int64), orr.to(torch.int64)
    ref = ol if ol is not None else orr
    return ol, orr, L.index_dtype_code(ref), ref.numel() - 1


def concat_2d_jagged(values_left, values_right, offsets_left, offsets_right, max_len_left, max_len_right,
                     max_seq_len, n_prefix=0) -> torch.Tensor:
    L.require_gpu_tensor(values_left, "values_left")
    L.require_gpu_tensor(values_right, "values_right")
    vl, vr = values_left.contiguous(), values_right.contiguous()
    dim = vl.shape[1]
    out = torch.empty((vl.shape[0] + vr.shape[0], dim), dtype=vl.dtype, device=vl.device)
    if out.shape[0] == 0:
        return out
    if offsets_left is None and offsets_right is None:
        batch, idt, ol, orr = vl.shape[0] // max_len_left, 0, None, None
    else:
        ol, orr, idt, batch = _pair_offsets(offsets_left, offsets_right)
        if DEBUG_CHECKS:
            _check_pair_lengths(ol, orr, max_len_left, max_len_right, max_seq_len, batch)
    with torch.cuda.device(vl.device):
        L.check(L.lib().hstu_concat_2d_jagged(vl.data_ptr(), vr.data_ptr(), out.data_ptr(), _vp(ol), _vp(orr),
                                              int(max_len_left or 0), int(max_len_right or 0), int(max_seq_len),
                                              batch, dim, vl.element_size(), int(n_prefix), idt,
                                              L.current_stream_ptr(vl.device)))
    return out


def split_2d_jagged(values, total_left, total_right, offsets_left, offsets_right, max_len_left, max_len_right,
                    max_seq_len, n_prefix=0) -> Tuple[torch.Tensor, torch.Tensor]:
    L.require_gpu_tensor(values, "values")
    vals = values.contiguous()
    dim = vals.shape[1]
    left = torch.empty((total_left, dim), dtype=vals.dtype, device=vals.device)
    right = torch.empty((total_right, dim), dtype=vals.dtype, device=vals.device)
    if vals.shape[0] == 0:
        return left, right
    ol, orr, idt, batch = _pair_offsets(offsets_left, offsets_right)
    if DEBUG_CHECKS:
        _check_pair_lengths(ol, orr, max_len_left, max_len_right, max_seq_len, batch)
    with torch.cuda.device(vals.device):
        L.check(L.lib().hstu_split_2d_jagged(vals.data_ptr(), left.data_ptr(), right.data_ptr(), _vp(ol), _vp(orr),
                                             int(max_len_left or 0), int(max_len_right or 0), int(max_seq_len),
                                             batch, dim, vals.element_size(), int(n_prefix), idt,
                                             L.current_stream_ptr(vals.device)))
    return left, right


def jagged_to_padded_dense(values, offsets, max_len) -> torch.Tensor:
    L.require_gpu_tensor(values, "values")
    vals = values.contiguous()
    offsets = _idx(offsets)
    B = offsets.numel() - 1
    dim = 1
    for s_ in vals.shape[1:]:
        dim *= s_
    dense = torch.empty((B, max_len) + tuple(vals.shape[1:]), dtype=vals.dtype, device=vals.device)
    if dense.numel() == 0:
        return dense
    with torch.cuda.device(vals.device):
        L.check(L.lib().hstu_jagged_to_padded_dense(vals.data_ptr(), dense.data_ptr(), offsets.data_ptr(), B,
                                                    int(max_len), dim, vals.element_size(),
                                                    L.index_dtype_code(offsets), L.current_stream_ptr(vals.device)))
    return dense


def jagged_write_tail_(values: torch.Tensor, dense: torch.Tensor, offsets: torch.Tensor, tail: int) -> torch.Tensor:
    """IN PLACE: the last ``tail`` rows of every user's region of ``values`` (sum L, ...) <- dense (B * tail, ...)."""
    L.require_gpu_tensor(values, "values")
    L.require_gpu_tensor(dense, "dense")
    if not values.is_contiguous():
        raise RuntimeError("jagged_write_tail_: the destination must be contiguous (it is written in place)")
    dense = dense.contiguous()
    offsets = _idx(offsets)
    B = offsets.shape[0] - 1
    dim = values[0].numel() if values.shape[0] else 0
    if dense.dtype != values.dtype or dense.shape[0] != B * tail or (dense.shape[0] and dense[0].numel() != dim):
        raise RuntimeError(f"jagged_write_tail_: dense {tuple(dense.shape)} {dense.dtype} does not hold {B} x {tail} rows of values {tuple(values.shape)} {values.dtype}")
    if B == 0 or tail == 0 or dim == 0:
        return values
    with torch.cuda.device(values.device):
        L.check(L.lib().hstu_jagged_write_tail(dense.data_ptr(), values.data_ptr(), offsets.data_ptr(), B, int(tail), dim,
                                               values.element_size(), L.index_dtype_code(offsets),
                                               L.current_stream_ptr(values.device)))
    return values


def dense_to_jagged(dense, offsets, total_rows) -> torch.Tensor:
    L.require_gpu_tensor(dense, "dense")
    d = dense.contiguous()
    offsets = _idx(offsets)
    B, max_len = d.shape[0], d.shape[1]
    dim = 1
    for s in d.shape[2:]:
        dim *= s
    out = torch.zeros((total_rows,) + tuple(d.shape[2:]), dtype=d.dtype, device=d.device)
    if out.numel() == 0 or d.numel() == 0:
        return out
    with torch.cuda.device(d.device):
        L.check(L.lib().hstu_dense_to_jagged(d.data_ptr(), out.data_ptr(), offsets.data_ptr(), B, max_len, dim,
                                             d.element_size(), L.index_dtype_code(offsets),
                                             L.current_stream_ptr(d.device)))
    return out


def expand_1d_jagged_to_dense(values, offsets, max_len) -> torch.Tensor:
    L.require_gpu_tensor(values, "values")
    vals, offsets = values.contiguous(), _idx(offsets)
    B = offsets.numel() - 1
    out = torch.empty((B, max_len), dtype=vals.dtype, device=vals.device)
    with torch.cuda.device(vals.device):
        L.check(L.lib().hstu_expand_1d_jagged_to_dense(vals.data_ptr(), offsets.data_ptr(), out.data_ptr(), B,
                                                       int(max_len), vals.element_size(),
                                                       L.index_dtype_code(offsets), L.current_stream_ptr(vals.device)))
    return out


def concat_1d_jagged_jagged(lengths_left, values_left, lengths_right, values_right) -> torch.Tensor:
    L.require_gpu_tensor(values_left, "values_left")
    ol = complete_cumsum(lengths_left.to(torch.int64))
    orr = complete_cumsum(lengths_right.to(torch.int64))
    vl, vr = values_left.contiguous(), values_right.contiguous()
    out = torch.empty(vl.numel() + vr.numel(), dtype=vl.dtype, device=vl.device)
    with torch.cuda.device(vl.device):
        L.check(L.lib().hstu_concat_1d_jagged_jagged(vl.data_ptr(), ol.data_ptr(), vr.data_ptr(), orr.data_ptr(),
                                                     out.data_ptr(), lengths_left.numel(), vl.element_size(),
                                                     L.HSTU_INDEX_I64, L.current_stream_ptr(vl.device)))
    return out


# ----------------------------------------------------------------------------- norms
def _f32(n, device):
    return torch.empty(n, dtype=torch.float32, device=device)


def layer_norm_fwd(x, weight, bias, eps):
    L.require_gpu_tensor(x, "x")
    x = x.contiguous()
    rows, dim = x.shape
    y = torch.empty_like(x)
    mean, rstd = _f32(rows, x.device), _f32(rows, x.device)
    w, b = weight.to(x.dtype).contiguous(), bias.to(x.dtype).contiguous()
    with torch.cuda.device(x.device):
        L.check(L.lib().hstu_layer_norm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                            rstd.data_ptr(), rows, dim, float(eps), L.torch_dtype_code(x.dtype),
                                            L.current_stream_ptr(x.device)))
    return y, mean, rstd


def swish_layer_norm_fwd(x, weight, bias, eps):
    """y = x * sigmoid(LayerNorm(x)) (hstu_swish_layer_norm_fwd): returns (y, mean, rstd)"""
    L.require_gpu_tensor(x, "x")
    x = x.contiguous()
    rows, dim = x.shape
    y = torch.empty_like(x)
    mean, rstd = _f32(rows, x.device), _f32(rows, x.device)
    w, b = weight.to(x.dtype).contiguous(), bias.to(x.dtype).contiguous()
    with torch.cuda.device(x.device):
        L.check(L.lib().hstu_swish_layer_norm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                  rstd.data_ptr(), rows, dim, float(eps), L.torch_dtype_code(x.dtype),
                                                  L.current_stream_ptr(x.device)))
    return y, mean, rstd


def swish_layer_norm_bwd(dy, x, weight, bias, mean, rstd):
    """dx, dweight (fp32), dbias (fp32) of swish_layer_norm_fwd"""
    dy, x = dy.contiguous(), x.contiguous()
    rows, dim = x.shape
    dx = torch.empty_like(x)
    dw, db = _f32(dim, x.device), _f32(dim, x.device)
    ws = torch.empty(L.lib().hstu_norm_bwd_workspace_bytes(rows, dim), dtype=torch.uint8, device=x.device)
    w, b = weight.to(x.dtype).contiguous(), bias.to(x.dtype).contiguous()
    with torch.cuda.device(x.device):
        L.check(L.lib().hstu_swish_layer_norm_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), b.data_ptr(), mean.data_ptr(),
                                                  rstd.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                                  rows, dim, L.torch_dtype_code(x.dtype), L.current_stream_ptr(x.device)))
    return dx, dw, db


def ln_linear_supported(x, n):
    """whether the fused LayerNorm + projection kernel takes (rows, k) activations and n output columns"""
    return bool(x.is_cuda and x.dim() == 2 and x.dtype in (torch.bfloat16, torch.float16) and
                L.lib().hstu_ln_linear_fwd_supported(x.shape[0], x.shape[1], n, L.torch_dtype_code(x.dtype)))


def ln_linear_fwd(x, ln_weight, ln_bias, eps, w_nk, bias, want_normed=False):
    """y = LayerNorm(x) @ w_nk.T + bias in one kernel (csrc/hstu_ln_linear.cuh): returns (y, normed_x or None, mean, rstd).
    ``w_nk``: the (n, k) K-contiguous weight in x's dtype."""
    L.require_gpu_tensor(x, "x")
    x = x.contiguous()
    rows, k = x.shape
    n = w_nk.shape[0]
    torch._assert(w_nk.shape[1] == k and w_nk.is_contiguous() and w_nk.dtype == x.dtype, "w_nk must be a contiguous (n, k) tensor of x's dtype")
    y = torch.empty((rows, n), dtype=x.dtype, device=x.device)
    normed = torch.empty_like(x) if want_normed else None
    mean, rstd = _f32(rows, x.device), _f32(rows, x.device)
    w, b = ln_weight.to(x.dtype).contiguous(), ln_bias.to(x.dtype).contiguous()
    pb = None if bias is None else bias.to(x.dtype).contiguous()
    with torch.cuda.device(x.device):
        L.check(L.lib().hstu_ln_linear_fwd(x.data_ptr(), k, w.data_ptr(), b.data_ptr(), float(eps), w_nk.data_ptr(),
                                           None if pb is None else pb.data_ptr(), y.data_ptr(), n,
                                           None if normed is None else normed.data_ptr(), k, mean.data_ptr(), rstd.data_ptr(),
                                           rows, k, n, L.torch_dtype_code(x.dtype), L.current_stream_ptr(x.device)))
    return y, normed, mean, rstd


def linear_k512_supported(x, n):
    """whether hstu_linear_k512 takes (rows, 512) activations (last dimension contiguous, row stride a multiple of 8) and n output columns"""
    return bool(x.is_cuda and x.dim() == 2 and x.dtype in (torch.bfloat16, torch.float16) and x.stride(1) == 1 and
                x.stride(0) % 8 == 0 and x.stride(0) >= x.shape[1] and x.data_ptr() % 16 == 0 and
                L.lib().hstu_linear_k512_supported(x.shape[0], x.shape[1], n, L.torch_dtype_code(x.dtype)))


def linear_k512(x, w_nk, bias=None):
    """y = x @ w_nk.T (+ bias) with a contraction length of 512 (csrc/hstu_ln_linear.cuh without the LayerNorm): the output
    stage's d y = d out . W_out^T, whose (n, k) operand is the layer's ``_output_weight`` as stored."""
    L.require_gpu_tensor(x, "x")
    rows, k = x.shape
    n = w_nk.shape[0]
    torch._assert(w_nk.shape[1] == k and w_nk.is_contiguous() and w_nk.dtype == x.dtype, "w_nk must be a contiguous (n, k) tensor of x's dtype")
    torch._assert(x.stride(1) == 1, "x must have a contiguous last dimension")
    y = torch.empty((rows, n), dtype=x.dtype, device=x.device)
    pb = None if bias is None else bias.to(x.dtype).contiguous()
    with torch.cuda.device(x.device):
        L.check(L.lib().hstu_linear_k512(x.data_ptr(), x.stride(0), w_nk.data_ptr(), None if pb is None else pb.data_ptr(),
                                         y.data_ptr(), n, rows, k, n, L.torch_dtype_code(x.dtype), L.current_stream_ptr(x.device)))
    return y


_ADDMM_LT = os.environ.get("HSTU_ADDMM_LT", "1") != "0"      # 0: torch.addmm (copy + in-place GEMM), the comparator
_ADDMM_WS: dict = {}          # device -> hipBLASLt workspace (32 MiB, as PyTorch's own), allocated once


def addmm_residual_supported(c, a, b) -> bool:
    """whether ``c + a @ b`` can run as ONE hipBLASLt launch with separate C and D buffers (hstu_addmm_residual): 2-D 16-bit CUDA
    operands of one dtype, c of the result's shape, rows contiguous and 16-byte aligned"""
    if not (_ADDMM_LT and a.is_cuda and a.dim() == 2 and b.dim() == 2 and c.dim() == 2 and a.dtype in (torch.bfloat16, torch.float16)
            and b.dtype == a.dtype and c.dtype == a.dtype and c.shape == (a.shape[0], b.shape[1]) and a.shape[1] == b.shape[0]):
        return False
    for t in (a, b, c):
        if t.stride(1) != 1 or t.stride(0) < t.shape[1] or t.stride(0) % 8 or t.data_ptr() % 16:
            return False
    return a.shape[0] > 0 and bool(L.lib().hstu_addmm_residual_supported())


def addmm_residual(c, a, b):
    """``c + a @ b`` (fp32 accumulation) without the copy of c that torch.addmm makes: hipBLASLt reads c and writes the result"""
    L.require_gpu_tensor(a, "a")
    m, k = a.shape
    n = b.shape[1]
    d = torch.empty((m, n), dtype=a.dtype, device=a.device)
    ws = _ADDMM_WS.get(a.device)
    if ws is None:
        ws = _ADDMM_WS[a.device] = torch.empty(32 << 20, dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        L.check(L.lib().hstu_addmm_residual(c.data_ptr(), c.stride(0), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), d.data_ptr(), n,
                                            m, n, k, L.torch_dtype_code(a.dtype), ws.data_ptr(), ws.numel(), L.current_stream_ptr(a.device)))
    return d


def layer_norm_bwd(dy, x, weight, mean, rstd, dresidual=None):
    """``dresidual``: a gradient that reaches x around the norm; added inside the kernel (dx = LN'(dy) + dresidual)."""
    dy, x = dy.contiguous(), x.contiguous()
    dres = None if dresidual is None else dresidual.contiguous()
    rows, dim = x.shape
    dx = torch.empty_like(x)
    dw, db = _f32(dim, x.device), _f32(dim, x.device)
    ws = torch.empty(L.lib().hstu_norm_bwd_workspace_bytes(rows, dim), dtype=torch.uint8, device=x.device)
    w = weight.to(x.dtype).contiguous()
    with torch.cuda.device(x.device):
        L.check(L.lib().hstu_layer_norm_bwd_residual(dy.data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr(),
                                                     rstd.data_ptr(), None if dres is None else dres.data_ptr(),
                                                     dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), rows, dim,
                                                     L.torch_dtype_code(x.dtype), L.current_stream_ptr(x.device)))
    return dx, dw, db


def _rows_view(t: torch.Tensor) -> torch.Tensor:
    """a 2-D tensor whose rows are contiguous (a column slice of a wider buffer stays a view)"""
    return t if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] else t.contiguous()


def norm_mul_fwd(attn, u, weight, bias, eps, num_heads, head_dim, group_norm, concat_ux, dropout_ratio=0.0, seed=0,
                 u_is_preactivation=False):
    """``dropout_ratio > 0``: the fused output-stage dropout (hstu_norm_mul_dropout_fwd); the mask is a function of
    (seed, element index) -- pass the same seed to norm_mul_bwd (and to a recompute of y).
    ``u_is_preactivation``: u is what goes INTO SiLU (e.g. the u slice of the uvqk buffer, read in place through its row
    stride); the kernel applies SiLU on the fly (hstu_norm_mul_silu_fwd)."""
    L.require_gpu_tensor(attn, "attn")
    attn, u = attn.contiguous(), _rows_view(u)
    rows, dim = attn.shape
    y = torch.empty((rows, 3 * dim if concat_ux else dim), dtype=attn.dtype, device=attn.device)
    ng = num_heads if group_norm else 1
    mean, rstd = _f32(rows * ng, attn.device), _f32(rows * ng, attn.device)
    w, b = weight.to(attn.dtype).contiguous(), bias.to(attn.dtype).contiguous()
    with torch.cuda.device(attn.device):
        L.check(L.lib().hstu_norm_mul_silu_fwd(attn.data_ptr(), u.data_ptr(), u.stride(0), int(bool(u_is_preactivation)),
                                               w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows,
                                               num_heads, head_dim, float(eps), int(group_norm), int(concat_ux),
                                               float(dropout_ratio), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                               L.torch_dtype_code(attn.dtype), L.current_stream_ptr(attn.device)))
    return y, mean, rstd


def norm_mul_bwd(dy, attn, u, weight, bias, mean, rstd, num_heads, head_dim, group_norm, concat_ux, dropout_ratio=0.0,
                 seed=0, u_is_preactivation=False, du=None):
    """``u_is_preactivation``: as in norm_mul_fwd; the returned du is then the gradient of the PRE-activation
    (d u * SiLU').  ``du``: where to write it (a column slice of a wider buffer is fine: the u slice of d uvqk)."""
    dy, attn, u = dy.contiguous(), attn.contiguous(), _rows_view(u)
    rows, dim = attn.shape
    dattn = torch.empty_like(attn)
    if du is None:
        du = torch.empty((rows, dim), dtype=attn.dtype, device=attn.device)
    assert du.shape == (rows, dim) and du.stride(1) == 1 and du.dtype == attn.dtype
    width = num_heads if group_norm else dim
    dw, db = _f32(width, attn.device), _f32(width, attn.device)
    ws = torch.empty(L.lib().hstu_norm_bwd_workspace_bytes(rows, dim), dtype=torch.uint8, device=attn.device)
    w, b = weight.to(attn.dtype).contiguous(), bias.to(attn.dtype).contiguous()
    with torch.cuda.device(attn.device):
        L.check(L.lib().hstu_norm_mul_silu_bwd(dy.data_ptr(), attn.data_ptr(), u.data_ptr(), u.stride(0),
                                               int(bool(u_is_preactivation)), w.data_ptr(), b.data_ptr(), mean.data_ptr(),
                                               rstd.data_ptr(), dattn.data_ptr(), du.data_ptr(), du.stride(0), dw.data_ptr(),
                                               db.data_ptr(), ws.data_ptr(), rows, num_heads, head_dim, int(group_norm),
                                               int(concat_ux), float(dropout_ratio), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                               L.torch_dtype_code(attn.dtype), L.current_stream_ptr(attn.device)))
    return dattn, du, dw, db


def silu_fwd(x: torch.Tensor) -> torch.Tensor:
    """silu over a 2-D (possibly column-sliced) tensor -> new contiguous tensor."""
    L.require_gpu_tensor(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib().hstu_silu_fwd(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], x.stride(0),
                                      out.stride(0), L.torch_dtype_code(x.dtype), L.current_stream_ptr(x.device)))
    return out


def silu_bwd(dout: torch.Tensor, x: torch.Tensor, din: Optional[torch.Tensor] = None) -> torch.Tensor:
    """din = dout * silu'(x); ``din`` may be a column slice of a larger buffer."""
    dout = dout if dout.stride(1) == 1 else dout.contiguous()
    assert x.stride(1) == 1
    if din is None:
        din = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib().hstu_silu_bwd(dout.data_ptr(), x.data_ptr(), din.data_ptr(), x.shape[0], x.shape[1],
                                      dout.stride(0), x.stride(0), din.stride(0), L.torch_dtype_code(x.dtype),
                                      L.current_stream_ptr(x.device)))
    return din


# ----------------------------------------------------------------------------- glue around the projections (ABI v10)
def cast_params(tensors, dtype, transpose_index: Optional[int] = None):
    """fp32 CUDA parameters -> ``dtype`` (bf16 / fp16) copies in ONE launch (hstu_cast_params); the tensor at
    ``transpose_index`` (2-D) comes back as its transposed, contiguous (cols, rows) copy.  All copies are views of one flat
    buffer (each starts 16-byte aligned)."""
    n = len(tensors)
    torch._assert(0 < n <= L.CAST_MAX_ITEMS, "cast_params: 1..8 tensors")
    dev = tensors[0].device
    srcs = []
    sizes = []
    for t in tensors:
        L.require_gpu_tensor(t, "parameter")
        torch._assert(t.dtype == torch.float32 and t.device == dev, "cast_params: fp32 tensors on one device")
        srcs.append(t.detach().contiguous())
        sizes.append((t.numel() + 7) // 8 * 8)
    flat = torch.empty(sum(sizes), dtype=dtype, device=dev)
    items = (L.HstuCastItem * n)()
    outs = []
    pos = 0
    for i, (t, sz) in enumerate(zip(srcs, sizes)):
        dst = flat[pos : pos + t.numel()]
        pos += sz
        items[i].src, items[i].dst, items[i].numel = t.data_ptr(), dst.data_ptr(), t.numel()
        if i == transpose_index:
            torch._assert(t.dim() == 2, "cast_params: the transposed item must be 2-D")
            items[i].rows, items[i].cols, items[i].transpose = t.shape[0], t.shape[1], 1
            outs.append(dst.view(t.shape[1], t.shape[0]))
        else:
            items[i].rows = items[i].cols = items[i].transpose = 0
            outs.append(dst.view(t.shape))
    with torch.cuda.device(dev):
        L.check(L.lib().hstu_cast_params(items, n, L.torch_dtype_code(dtype), L.current_stream_ptr(dev)))
    return outs


def column_sum(x: torch.Tensor, out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 column sums of a 2-D bf16 / fp16 tensor with contiguous rows (hstu_column_sum: fixed summation order)"""
    L.require_gpu_tensor(x, "x")
    torch._assert(x.dim() == 2 and x.stride(1) == 1, "column_sum: a 2-D tensor with contiguous rows")
    rows, cols = x.shape
    if out is None:
        out = _f32(cols, x.device)
    if workspace is None:
        workspace = torch.empty(max(16, L.lib().hstu_column_sum_workspace_bytes(rows, cols)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        L.check(L.lib().hstu_column_sum(x.data_ptr(), x.stride(0) if rows > 1 else max(cols, x.stride(0)), rows, cols, out.data_ptr(),
                                        workspace.data_ptr(), L.torch_dtype_code(x.dtype), L.current_stream_ptr(x.device)))
    return out


def column_sum_supported(x: torch.Tensor) -> bool:
    # (stride(0) >= columns: an expanded / broadcast gradient -- row stride 0 -- is torch's to sum)
    return bool(x.is_cuda and x.dim() == 2 and x.dtype in (torch.bfloat16, torch.float16) and x.stride(1) == 1 and
                x.shape[1] % 8 == 0 and x.stride(0) % 8 == 0 and x.stride(0) >= x.shape[1] and x.data_ptr() % 16 == 0)


def calib_mfma_stream(device, iters: int = 4096):
    """(launch, flop per launch): the MFMA calibration stream of bench.py (hstu_calib_mfma_stream)"""
    sink = _f32(4, device)
    flops = C.c_double(0.0)

    def launch():
        with torch.cuda.device(device):
            L.check(L.lib().hstu_calib_mfma_stream(int(iters), sink.data_ptr(), C.byref(flops), L.current_stream_ptr(device)))

    launch()
    return launch, flops.value


def calib_read_stream(src: torch.Tensor):
    """launch(): one non-temporal read of ``src`` (hstu_calib_read_stream)"""
    sink = _f32(4, src.device)
    nbytes = src.numel() * src.element_size()

    def launch():
        with torch.cuda.device(src.device):
            L.check(L.lib().hstu_calib_read_stream(src.data_ptr(), nbytes, sink.data_ptr(), L.current_stream_ptr(src.device)))

    return launch


# ----------------------------------------------------------------------------- row L2 normalisation
def l2_norm_fwd(x: torch.Tensor, eps: float) -> torch.Tensor:
    L.require_gpu_tensor(x, "x")
    x = x.contiguous()
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1] if x.numel() else 0
    with torch.cuda.device(x.device):
        L.check(L.lib().hstu_l2_norm_fwd(x.data_ptr(), y.data_ptr(), rows, x.shape[-1], float(eps), L.torch_dtype_code(x.dtype),
                                         L.current_stream_ptr(x.device)))
    return y


def l2_norm_bwd(dy: torch.Tensor, x: torch.Tensor, eps: float) -> torch.Tensor:
    dy, x = dy.contiguous(), x.contiguous()
    dx = torch.empty_like(x)
    rows = x.numel() // x.shape[-1] if x.numel() else 0
    with torch.cuda.device(x.device):
        L.check(L.lib().hstu_l2_norm_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), rows, x.shape[-1], float(eps),
                                         L.torch_dtype_code(x.dtype), L.current_stream_ptr(x.device)))
    return dx


# ----------------------------------------------------------------------------- sampled-softmax loss
def _ss_rows(t: torch.Tensor, name: str) -> torch.Tensor:
    L.require_gpu_tensor(t, name)
    if t.dim() != 2:
        raise RuntimeError(f"{name} must be 2-D, got {tuple(t.shape)}")
    return t if t.stride(1) == 1 else t.contiguous()


def sampled_softmax_fwd(q, pos_emb, pos_ids, neg_rows, neg_ids, table, temperature, pos_l2_norm, table_l2_norm, eps):
    """-> (row_loss, lse), fp32 (n_rows)."""
    q, pos_emb, table = _ss_rows(q, "output_embeddings"), _ss_rows(pos_emb, "supervision_embeddings"), _ss_rows(table, "table")
    if not (q.dtype == pos_emb.dtype == table.dtype):
        raise RuntimeError(f"embeddings must share one dtype, got {q.dtype}, {pos_emb.dtype}, {table.dtype}")
    n, D = q.shape
    neg_rows, neg_ids, pos_ids = neg_rows.contiguous(), neg_ids.contiguous(), pos_ids.contiguous()
    if neg_rows.dtype != torch.int64 or neg_ids.dtype != torch.int64 or pos_ids.dtype != torch.int64:
        raise RuntimeError("ids must be int64")
    if neg_rows.shape != neg_ids.shape or neg_rows.dim() != 2 or neg_rows.shape[0] != n or pos_ids.shape != (n,):
        raise RuntimeError("sampled ids must be (rows, num_negatives), supervision ids (rows,)")
    row_loss = torch.empty(n, dtype=torch.float32, device=q.device)
    lse = torch.empty_like(row_loss)
    with torch.cuda.device(q.device):
        L.check(L.lib().hstu_sampled_softmax_fwd(
            q.data_ptr(), q.stride(0), pos_emb.data_ptr(), pos_emb.stride(0), pos_ids.data_ptr(), neg_rows.data_ptr(),
            neg_ids.data_ptr(), table.data_ptr(), table.stride(0), table.shape[0], n, neg_rows.shape[1], D, float(temperature),
            int(pos_l2_norm), int(table_l2_norm), float(eps), row_loss.data_ptr(), lse.data_ptr(), L.torch_dtype_code(q.dtype),
            L.current_stream_ptr(q.device)))
    return row_loss, lse


def sampled_softmax_bwd(g_row, lse, q, pos_emb, pos_ids, neg_rows, neg_ids, table, temperature, pos_l2_norm, table_l2_norm, eps):
    """-> (dq, dpos_emb) in the embedding dtype, dtable fp32 (table_rows, dim)."""
    q, pos_emb, table = _ss_rows(q, "output_embeddings"), _ss_rows(pos_emb, "supervision_embeddings"), _ss_rows(table, "table")
    n, D = q.shape
    neg_rows, neg_ids, pos_ids = neg_rows.contiguous(), neg_ids.contiguous(), pos_ids.contiguous()
    g_row = g_row.to(torch.float32).contiguous()
    dq, dpos = torch.empty_like(q), torch.empty_like(pos_emb)
    dtable = torch.zeros(table.shape[0], D, dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        L.check(L.lib().hstu_sampled_softmax_bwd(
            q.data_ptr(), q.stride(0), pos_emb.data_ptr(), pos_emb.stride(0), pos_ids.data_ptr(), neg_rows.data_ptr(),
            neg_ids.data_ptr(), table.data_ptr(), table.stride(0), table.shape[0], n, neg_rows.shape[1], D, float(temperature),
            int(pos_l2_norm), int(table_l2_norm), float(eps), lse.data_ptr(), g_row.data_ptr(), dq.data_ptr(), dq.stride(0),
            dpos.data_ptr(), dpos.stride(0), dtable.data_ptr(), L.torch_dtype_code(q.dtype), L.current_stream_ptr(q.device)))
    return dq, dpos, dtable

