"""Drop-in for the research-path HSTU layer of
generative_recommenders/research/modeling/sequential/hstu.py: the relative attention bias
modules (:51-144), the attention (:150-223) and ``SequentialTransductionUnitJagged`` (:226-444),
on the fused HIP kernels: the (B, N, N) bias and the (B, H, N, N) logits of the reference are
never materialised -- the bias is generated inside the attention kernels from
``(pos_w, ts_w, timestamps)`` and its gradient is reduced per workgroup.

Parameter names match the reference (``_uvqk``, ``_o.weight``, ``_o.bias``,
``_rel_attn_bias._ts_w`` / ``_pos_w`` / ``_w``) so its state_dicts load unchanged.
The incremental branch (``delta_x_offsets`` / ``cache``, :160-191, 318-336, 421-429) runs on the delta-q form
of the same kernels: the new row's attention reads the cached K (with the new row inserted) and V directly, the
(B, N, N) product of all rows the reference recomputes and then discards is never formed.
Not supported: the ``softmax_rel_bias`` normalisation (no shipped config uses it).
"""

import abc
import ctypes as C
import math
from typing import Callable, Dict, List, Optional, Tuple, Union

import torch
import torch.nn.functional as F

from generative_recommenders_amd import _lib as L
from generative_recommenders_amd.ops import _launch
from generative_recommenders_amd.ops.hstu_attention import _pad_head_dim


class RelativeAttentionBiasModule(torch.nn.Module):
    """Parameter holder; the bias itself is produced inside the attention kernel."""

    @abc.abstractmethod
    def bias_params(self) -> Tuple[torch.Tensor, Optional[torch.Tensor], int, float]:
        """(pos_w, ts_w or None, num_buckets, bucket_div)"""


class RelativePositionalBias(RelativeAttentionBiasModule):
    def __init__(self, max_seq_len: int) -> None:
        super().__init__()
        self._max_seq_len = max_seq_len
        self._w = torch.nn.Parameter(torch.empty(2 * max_seq_len - 1).normal_(mean=0, std=0.02))

    def bias_params(self):
        return self._w, None, 0, 1.0


class RelativeBucketedTimeAndPositionBasedBias(RelativeAttentionBiasModule):
    """Bucketizes timespans based on ts(next-item) - ts(current-item).  ``bucketization_fn`` is kept
    for signature compatibility; the kernels implement the function every shipped config uses,
    ``floor(log(max(|x|, 1)) / bucket_div)`` with ``bucket_div = 0.301``."""

    def __init__(self, max_seq_len: int, num_buckets: int,
                 bucketization_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                 bucket_div: float = 0.301) -> None:
        super().__init__()
        self._max_seq_len = max_seq_len
        self._ts_w = torch.nn.Parameter(torch.empty(num_buckets + 1).normal_(mean=0, std=0.02))
        self._pos_w = torch.nn.Parameter(torch.empty(2 * max_seq_len - 1).normal_(mean=0, std=0.02))
        self._num_buckets = num_buckets
        self._bucketization_fn = bucketization_fn
        self._bucket_div = bucket_div

    def bias_params(self):
        return self._pos_w, self._ts_w, self._num_buckets, self._bucket_div


def _fill_bias(p: L.HstuAttnParams, pos_w, ts_w, timestamps, num_buckets, bucket_div):
    p.pos_w = pos_w.data_ptr()
    if ts_w is not None:
        p.ts_w = ts_w.data_ptr()
        p.timestamps = timestamps.data_ptr()
        p.ts_row_stride = timestamps.stride(0)
    p.num_buckets = int(num_buckets)
    p.bucket_div = float(bucket_div)


_ORDER_MIN_USERS = 512       # batches from this size on are launched heavy users first


class _RelBiasAttentionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, q, k, v, x_offsets, timestamps, pos_w, ts_w, num_buckets, bucket_div):
        for name, t in (("q", q), ("k", k), ("v", v), ("x_offsets", x_offsets), ("pos_w", pos_w)):
            L.require_gpu_tensor(t, name)
        q, k, v = _launch._aligned_rows(q), _launch._aligned_rows(k), _launch._aligned_rows(v)
        # heavy users first (the reference's sort_by_length, ops/triton/triton_hstu_attention.py:1968-1973): the persistent backward
        # kernels hand users out from a counter, and a launch that ends on its longest users waits for them alone (ML-20M lengths,
        # 8192 users: backward 2.46 -> 2.35 ms).  One argsort per BATCH: the layers of a model pass the same offsets tensor object
        # (_launch.length_order keeps the last result); results never depend on the order.
        order = _launch.length_order(x_offsets) if x_offsets.numel() > _ORDER_MIN_USERS else None
        x_offsets = _launch._idx(x_offsets)
        pos32 = pos_w.detach().float().contiguous()
        ts32 = None if ts_w is None else ts_w.detach().float().contiguous()
        ts = None if ts_w is None else timestamps.to(torch.int64).contiguous()
        torch._assert(pos32.numel() == 2 * n - 1, "pos_w must have 2 * n - 1 entries")
        out = torch.empty((q.shape[0], q.shape[1], v.shape[2]), dtype=q.dtype, device=q.device)
        if q.shape[0]:
            p = L.HstuAttnParams()
            _launch._fill_attn_params(p, q, k, v, out, x_offsets, None, n, 1.0, 1.0 / n, 0, 0, 0, 0, user_order=order)
            _fill_bias(p, pos32, ts32, ts, num_buckets, bucket_div)
            with torch.cuda.device(q.device):
                L.check(L.lib().hstu_attn_fwd(C.byref(p), L.current_stream_ptr(q.device)))
        ctx.user_order = order
        ctx.save_for_backward(q, k, v, x_offsets, pos32, *([ts32, ts] if ts32 is not None else []))
        ctx.has_ts = ts32 is not None
        ctx.meta = (n, num_buckets, bucket_div, pos_w.dtype, None if ts_w is None else ts_w.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, x_offsets, pos32 = ctx.saved_tensors[:5]
        ts32, ts = (ctx.saved_tensors[5], ctx.saved_tensors[6]) if ctx.has_ts else (None, None)
        n, num_buckets, bucket_div, pos_dtype, ts_dtype = ctx.meta
        dout = _launch._aligned_rows(dout)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dpos = torch.zeros_like(pos32)
        dts = None if ts32 is None else torch.zeros_like(ts32)
        if q.shape[0]:
            bp = L.HstuAttnBwdParams()
            _launch._fill_attn_params(bp.fwd, q, k, v, None, x_offsets, None, n, 1.0, 1.0 / n, 0, 0, 0, 0, user_order=ctx.user_order)
            _fill_bias(bp.fwd, pos32, ts32, ts, num_buckets, bucket_div)
            bp.dout, bp.dq, bp.dk, bp.dv = dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
            bp.do_row_stride, bp.do_head_stride = dout.stride(0), dout.stride(1)
            bp.dq_row_stride, bp.dq_head_stride = dq.stride(0), dq.stride(1)
            bp.dk_row_stride, bp.dk_head_stride = dk.stride(0), dk.stride(1)
            bp.dv_row_stride, bp.dv_head_stride = dv.stride(0), dv.stride(1)
            bp.total_rows = q.shape[0]
            bp.dpos_w = dpos.data_ptr()
            bp.dts_w = None if dts is None else dts.data_ptr()
            ws_bytes = L.lib().hstu_attn_bwd_workspace_bytes(C.byref(bp))
            ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=q.device)
            bp.workspace = ws.data_ptr()
            with torch.cuda.device(q.device):
                L.check(L.lib().hstu_attn_bwd(C.byref(bp), L.current_stream_ptr(q.device)))
        return (None, dq, dk, dv, None, None, dpos.to(pos_dtype), None if dts is None else dts.to(ts_dtype), None, None)


def _heads3(q, k, v, num_heads, attention_dim, linear_dim):
    """(rows, H*d) -> (rows, H, d padded): 16-byte vectors, and one head dim for q/k and v (the bias kernels are
    instantiated for dqk == dv: the smaller one is zero-padded, which changes neither q.k nor the sliced output)"""
    q3 = _pad_head_dim(q.view(q.shape[0], num_heads, attention_dim))
    k3 = _pad_head_dim(k.view(k.shape[0], num_heads, attention_dim))
    v3 = _pad_head_dim(v.view(v.shape[0], num_heads, linear_dim))
    dpad = max(q3.shape[2], v3.shape[2])
    if q3.shape[2] != dpad:
        q3, k3 = F.pad(q3, (0, dpad - q3.shape[2])), F.pad(k3, (0, dpad - k3.shape[2]))
    if v3.shape[2] != dpad:
        v3 = F.pad(v3, (0, dpad - v3.shape[2]))
    return q3, k3, v3


def hstu_rel_bias_attention(num_heads: int, attention_dim: int, linear_dim: int, q: torch.Tensor, k: torch.Tensor,
                            v: torch.Tensor, x_offsets: torch.Tensor, all_timestamps: Optional[torch.Tensor], n: int,
                            rel_attn_bias: RelativeAttentionBiasModule) -> torch.Tensor:
    """Fused equivalent of ``_hstu_attention_maybe_from_cache`` (hstu.py:150-223) for the full-sequence
    case: q, k (sum L, H*attention_dim), v (sum L, H*linear_dim) -> (sum L, H*linear_dim);
    P = silu(q.k + rel_bias) / n with the lower-triangular (diagonal included) mask.  Without timestamps the reference
    adds NO bias at all, positional term included (``if all_timestamps is not None``, :205-206): that case is the
    plain causal attention of the ops path with alpha = 1."""
    L_ = q.shape[0]
    q3, k3, v3 = _heads3(q, k, v, num_heads, attention_dim, linear_dim)
    if all_timestamps is None:
        from generative_recommenders_amd.ops.hstu_attention import hip_hstu_mha

        out = hip_hstu_mha(n, 1.0, q3, k3, v3, x_offsets)
    else:
        pos_w, ts_w, nb, div = rel_attn_bias.bias_params()
        out = _RelBiasAttentionFunction.apply(n, q3, k3, v3, x_offsets, all_timestamps, pos_w, ts_w, nb, div)
    return out[..., :linear_dim].reshape(L_, num_heads * linear_dim)


def hstu_rel_bias_delta_attention(num_heads: int, attention_dim: int, linear_dim: int, delta_q: torch.Tensor,
                                  k: torch.Tensor, v: torch.Tensor, x_offsets: torch.Tensor,
                                  all_timestamps: Optional[torch.Tensor], n: int,
                                  rel_attn_bias: RelativeAttentionBiasModule) -> torch.Tensor:
    """Attention of every user's LAST row only (forward only): delta_q (B, H*attention_dim) is the query of position
    len_b - 1, k / v are the full jagged tensors with that row already in place.  What the reference's incremental
    branch needs from its (B, H, N, N) product (hstu.py:160-223, then ``attn_output[delta_x_offsets[0]]`` :421-425)."""
    B = x_offsets.numel() - 1
    q3, k3, v3 = _heads3(delta_q, k, v, num_heads, attention_dim, linear_dim)
    for name, t in (("delta_q", q3), ("k", k3), ("v", v3), ("x_offsets", x_offsets)):
        L.require_gpu_tensor(t, name)
    q3, k3, v3 = _launch._aligned_rows(q3), _launch._aligned_rows(k3), _launch._aligned_rows(v3)
    offs = _launch._idx(x_offsets)
    out = torch.empty((B, num_heads, v3.shape[2]), dtype=q3.dtype, device=q3.device)
    if B and k3.shape[0]:
        p = L.HstuAttnParams()
        _launch._fill_attn_params(p, q3, k3, v3, out, offs, None, n, 1.0, 1.0 / n, 0, 0, 0, 1)
        keep = []
        if all_timestamps is not None:
            pos_w, ts_w, nb, div = rel_attn_bias.bias_params()
            pos32 = pos_w.detach().float().contiguous()
            ts32 = None if ts_w is None else ts_w.detach().float().contiguous()
            ts = None if ts_w is None else all_timestamps.to(torch.int64).contiguous()
            torch._assert(pos32.numel() == 2 * n - 1, "pos_w must have 2 * n - 1 entries")
            _fill_bias(p, pos32, ts32, ts, nb, div)
            keep = [pos32, ts32, ts]
        with torch.cuda.device(q3.device):
            L.check(L.lib().hstu_attn_fwd(C.byref(p), L.current_stream_ptr(q3.device)))
        del keep
    return out[..., :linear_dim].reshape(B, num_heads * linear_dim)


HSTUCacheState = Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor], torch.Tensor]


class SequentialTransductionUnitJagged(torch.nn.Module):
    def __init__(self, embedding_dim: int, linear_hidden_dim: int, attention_dim: int, dropout_ratio: float,
                 attn_dropout_ratio: float, num_heads: int, linear_activation: str,
                 relative_attention_bias_module: Optional[RelativeAttentionBiasModule] = None,
                 normalization: str = "rel_bias", linear_config: str = "uvqk", concat_ua: bool = False,
                 epsilon: float = 1e-6, max_length: Optional[int] = None) -> None:
        super().__init__()
        self._embedding_dim = embedding_dim
        self._linear_dim = linear_hidden_dim
        self._attention_dim = attention_dim
        self._dropout_ratio = dropout_ratio
        self._attn_dropout_ratio = attn_dropout_ratio   # stored, never applied (as in the reference)
        self._num_heads = num_heads
        self._rel_attn_bias = relative_attention_bias_module
        if normalization not in ("rel_bias", "hstu_rel_bias"):
            raise ValueError(f"Unknown normalization method {normalization}")
        if linear_config != "uvqk":
            raise ValueError(f"Unknown linear_config {linear_config}")
        self._normalization = normalization
        self._linear_config = linear_config
        self._uvqk = torch.nn.Parameter(
            torch.empty((embedding_dim, linear_hidden_dim * 2 * num_heads + attention_dim * num_heads * 2)).normal_(
                mean=0, std=0.02))
        self._linear_activation = linear_activation
        self._concat_ua = concat_ua
        self._o = torch.nn.Linear(in_features=linear_hidden_dim * num_heads * (3 if concat_ua else 1),
                                  out_features=embedding_dim)
        torch.nn.init.xavier_uniform_(self._o.weight)
        self._eps = epsilon

    def _uvqk_rows(self, x: torch.Tensor):
        """LN (no affine) -> x W -> activation -> u, v, q, k for the rows of x (hstu.py:316-336)"""
        from generative_recommenders_amd.ops.hstu_compute import _SiluFunction
        from generative_recommenders_amd.ops.layer_norm import layer_norm

        D, H, Ld, A = self._embedding_dim, self._num_heads, self._linear_dim, self._attention_dim
        ones = torch.ones(D, dtype=x.dtype, device=x.device)
        normed_x = layer_norm(x, ones, torch.zeros_like(ones), self._eps)
        mm = torch.mm(normed_x, self._uvqk.to(x.dtype))
        if self._linear_activation == "silu":
            mm = _SiluFunction.apply(mm)                                            # SiLU on all of u, v, q, k
        elif self._linear_activation != "none":
            raise ValueError(f"Unknown linear_activation {self._linear_activation}")
        return torch.split(mm, [Ld * H, Ld * H, A * H, A * H], dim=1)

    def _output_rows(self, u: torch.Tensor, attn: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """o(dropout(u * LN(attn) | [u, LN(attn), u * LN(attn)])) + x for the rows given (hstu.py:426-444)"""
        from generative_recommenders_amd.ops.hstu_compute import _NormMulFunction, draw_dropout_seed
        from generative_recommenders_amd.ops.layer_norm import layer_norm

        H, Ld = self._num_heads, self._linear_dim
        w1 = torch.ones(Ld * H, dtype=x.dtype, device=x.device)
        if self._concat_ua:
            a = layer_norm(attn, w1, torch.zeros_like(w1), self._eps)
            o_input = F.dropout(torch.cat([u, a, u * a], dim=-1), p=self._dropout_ratio, training=self.training)
        else:
            # dropout inside the norm kernel (mask regenerated from the seed in backward: ops/hstu_compute.py)
            p_drop = float(self._dropout_ratio) if self.training else 0.0
            o_input = _NormMulFunction.apply(attn, u.contiguous(), w1, torch.zeros_like(w1), self._eps, H, Ld, False, False,
                                             p_drop, draw_dropout_seed() if p_drop > 0.0 else 0)
        return self._o(o_input) + x

    def forward(self, x: torch.Tensor, x_offsets: torch.Tensor, all_timestamps: Optional[torch.Tensor],
                invalid_attn_mask: torch.Tensor, delta_x_offsets: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                cache: Optional[HSTUCacheState] = None, return_cache_states: bool = False):
        """x (sum L, D) -> x' (sum L, D), and the cache state ``(v, padded_q, padded_k, x')``.
        ``invalid_attn_mask`` only supplies n: the kernels apply the lower-triangular mask the reference registers
        (hstu.py:626-638).  ``padded_q`` / ``padded_k`` (B, n, H*A) are built when ``return_cache_states`` is set or the
        call is incremental -- the reference materialises them in every call as a by-product of its padded attention;
        here they would be two extra passes nobody reads.

        Incremental call (``delta_x_offsets = (rows, cols)``, ``cache`` from an earlier call): as in the reference, only
        the rows ``rows`` of x are new; they are projected, written into the cached v / padded q / padded k / outputs
        IN PLACE (``index_copy_``), and their attention is computed against the cached keys.  ``cols[b]`` is the
        position of user b's new row; it is the user's last row in incremental decoding (``cols == lengths - 1``,
        delta-q kernel directly) -- for any other position the keys / values are first cut to ``cols[b] + 1`` rows."""
        assert self._rel_attn_bias is not None
        n = invalid_attn_mask.size(-1)
        H, Ld, A = self._num_heads, self._linear_dim, self._attention_dim
        if delta_x_offsets is None:
            u, v, q, k = self._uvqk_rows(x)
            attn = hstu_rel_bias_attention(H, A, Ld, q, k, v, x_offsets, all_timestamps, n, self._rel_attn_bias)
            new_outputs = self._output_rows(u, attn, x)
            padded_q = padded_k = None
            if return_cache_states:
                v = v.contiguous()
                padded_q = _launch.jagged_to_padded_dense(q.detach(), x_offsets, n)
                padded_k = _launch.jagged_to_padded_dense(k.detach(), x_offsets, n)
            return new_outputs, (v, padded_q, padded_k, new_outputs)

        assert cache is not None
        rows, cols = delta_x_offsets
        cached_v, cached_q, cached_k, cached_outputs = cache
        assert cached_q is not None and cached_k is not None, "the cache must come from a call with return_cache_states=True"
        B = x_offsets.size(0) - 1
        xd = x[rows, :]
        u, v, q, k = self._uvqk_rows(xd)
        v_full = cached_v.index_copy_(0, rows, v.to(cached_v.dtype))
        flat = cols + torch.arange(0, B * n, n, device=cols.device, dtype=cols.dtype)
        padded_q = cached_q.view(B * n, -1).index_copy_(0, flat, q.to(cached_q.dtype)).view(B, n, -1)
        padded_k = cached_k.view(B * n, -1).index_copy_(0, flat, k.to(cached_k.dtype)).view(B, n, -1)
        lengths = x_offsets[1:] - x_offsets[:-1]
        k_jag = _launch.dense_to_jagged(padded_k, x_offsets, cached_v.shape[0])
        if bool((cols.to(lengths.dtype) == lengths - 1).all()):
            kk, vv, offs = k_jag, v_full, x_offsets
        else:   # the new row is not the last one: keys / values of user b end at position cols[b]
            user = torch.repeat_interleave(torch.arange(B, device=x.device), lengths.to(torch.int64))
            pos = torch.arange(cached_v.shape[0], device=x.device) - x_offsets[:-1].to(torch.int64)[user]
            keep = pos <= cols.to(torch.int64)[user]
            kk, vv = k_jag[keep], v_full[keep]
            offs = torch.zeros(B + 1, dtype=x_offsets.dtype, device=x.device)
            offs[1:] = torch.cumsum(cols.to(x_offsets.dtype) + 1, 0)
        attn = hstu_rel_bias_delta_attention(H, A, Ld, q, kk, vv, offs, all_timestamps, n, self._rel_attn_bias)
        new_rows = self._output_rows(u, attn, xd)
        new_outputs = cached_outputs.index_copy_(0, rows, new_rows.to(cached_outputs.dtype))
        return new_outputs, (v_full, padded_q, padded_k, new_outputs)


class HSTUJagged(torch.nn.Module):
    """The layer stack (hstu.py:447-540): same constructor and methods.  ``autocast_dtype``: the reference wraps the
    stack in ``torch.autocast`` (matmuls in that dtype, norms in fp32); here the activations are CAST to that dtype at
    the entry of the stack and back at its exit -- the HIP ops compute in fp32 internally whatever the I/O dtype, and
    the parameters stay fp32 (``.to(x.dtype)`` per call)."""

    def __init__(self, modules: List[SequentialTransductionUnitJagged], autocast_dtype: Optional[torch.dtype]) -> None:
        super().__init__()
        self._attention_layers = torch.nn.ModuleList(modules=modules)
        self._autocast_dtype = autocast_dtype

    def jagged_forward(self, x: torch.Tensor, x_offsets: torch.Tensor, all_timestamps: Optional[torch.Tensor],
                       invalid_attn_mask: torch.Tensor, delta_x_offsets: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                       cache: Optional[List[HSTUCacheState]] = None, return_cache_states: bool = False
                       ) -> Tuple[torch.Tensor, List[HSTUCacheState]]:
        cache_states: List[HSTUCacheState] = []
        in_dtype = x.dtype
        if self._autocast_dtype is not None:
            x = x.to(self._autocast_dtype)
        for i, layer in enumerate(self._attention_layers):
            x, cache_states_i = layer(x=x, x_offsets=x_offsets, all_timestamps=all_timestamps,
                                      invalid_attn_mask=invalid_attn_mask, delta_x_offsets=delta_x_offsets,
                                      cache=cache[i] if cache is not None else None,
                                      return_cache_states=return_cache_states)
            if return_cache_states:
                cache_states.append(cache_states_i)
        return x.to(in_dtype), cache_states

    def forward(self, x: torch.Tensor, x_offsets: torch.Tensor, all_timestamps: Optional[torch.Tensor],
                invalid_attn_mask: torch.Tensor, delta_x_offsets: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                cache: Optional[List[HSTUCacheState]] = None, return_cache_states: bool = False
                ) -> Tuple[torch.Tensor, List[HSTUCacheState]]:
        """x (B, N, D) or jagged (sum L, D) -> (B, N, D) padded with zeros (hstu.py:503-540)"""
        if x.dim() == 3:
            total = int(x_offsets[-1].item())
            x = _DenseToJagged.apply(x, x_offsets, total)
        jagged_x, cache_states = self.jagged_forward(x=x, x_offsets=x_offsets, all_timestamps=all_timestamps,
                                                     invalid_attn_mask=invalid_attn_mask, delta_x_offsets=delta_x_offsets,
                                                     cache=cache, return_cache_states=return_cache_states)
        y = _JaggedToPadded.apply(jagged_x, x_offsets, invalid_attn_mask.size(1))
        return y, cache_states


class _DenseToJagged(torch.autograd.Function):
    """fbgemm.dense_to_jagged / jagged_to_padded_dense with gradients (each is the other's backward)"""

    @staticmethod
    def forward(ctx, dense, offsets, total):
        ctx.save_for_backward(offsets)
        ctx.n = dense.shape[1]
        return _launch.dense_to_jagged(dense, offsets, total)

    @staticmethod
    def backward(ctx, g):
        (offsets,) = ctx.saved_tensors
        return _launch.jagged_to_padded_dense(g.contiguous(), offsets, ctx.n), None, None


class _JaggedToPadded(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values, offsets, n):
        ctx.save_for_backward(offsets)
        ctx.total = values.shape[0]
        return _launch.jagged_to_padded_dense(values, offsets, n)

    @staticmethod
    def backward(ctx, g):
        (offsets,) = ctx.saved_tensors
        return _launch.dense_to_jagged(g.contiguous(), offsets, ctx.total), None, None


def get_current_embeddings(lengths: torch.Tensor, encoded_embeddings: torch.Tensor) -> torch.Tensor:
    """(B, N, D) -> (B, D): the row at position length - 1 (research/modeling/sequential/utils.py:74-90)"""
    B, N, D = encoded_embeddings.size()
    flat = (lengths - 1) + torch.arange(0, B * N, N, device=lengths.device, dtype=lengths.dtype)
    return encoded_embeddings.reshape(-1, D)[flat, :].reshape(B, -1)


class HSTU(torch.nn.Module):
    """The research-path model (hstu.py:543-809): embedding / preprocessor / postprocessor / similarity modules are the
    caller's (any objects with the reference's method names: ``item_embedding_dim``, ``get_item_embeddings``, the
    preprocessor's ``__call__(past_lengths, past_ids, past_embeddings, past_payloads)``, ...); what is built here is the
    ``HSTUJagged`` stack on the HIP kernels, with the reference's constructor arguments, attribute names (``_hstu``,
    ``_attn_mask``, ``_embedding_module`` ... -- state_dicts load unchanged) and methods.  ``similarity_fn`` follows
    ``SequentialEncoderWithLearnedSimilarityModule`` (research/modeling/similarity_module.py:23-67)."""

    def __init__(self, max_sequence_len: int, max_output_len: int, embedding_dim: int, num_blocks: int, num_heads: int,
                 linear_dim: int, attention_dim: int, normalization: str, linear_config: str, linear_activation: str,
                 linear_dropout_rate: float, attn_dropout_rate: float, embedding_module, similarity_module,
                 input_features_preproc_module, output_postproc_module, enable_relative_attention_bias: bool = True,
                 concat_ua: bool = False, verbose: bool = True) -> None:
        super().__init__()
        self._ndp_module = similarity_module
        self._embedding_dim = embedding_dim
        self._item_embedding_dim = embedding_module.item_embedding_dim
        self._max_sequence_length = max_sequence_len
        self._embedding_module = embedding_module
        self._input_features_preproc = input_features_preproc_module
        self._output_postproc = output_postproc_module
        self._num_blocks = num_blocks
        self._num_heads = num_heads
        self._dqk = attention_dim
        self._dv = linear_dim
        self._linear_activation = linear_activation
        self._linear_dropout_rate = linear_dropout_rate
        self._attn_dropout_rate = attn_dropout_rate
        self._enable_relative_attention_bias = enable_relative_attention_bias
        self._hstu = HSTUJagged(
            modules=[SequentialTransductionUnitJagged(
                embedding_dim=embedding_dim, linear_hidden_dim=linear_dim, attention_dim=attention_dim,
                normalization=normalization, linear_config=linear_config, linear_activation=linear_activation,
                num_heads=num_heads,
                relative_attention_bias_module=(RelativeBucketedTimeAndPositionBasedBias(
                    max_seq_len=max_sequence_len + max_output_len, num_buckets=128)
                    if enable_relative_attention_bias else None),   # None: forward asserts, as the reference's does (:342)
                dropout_ratio=linear_dropout_rate, attn_dropout_ratio=attn_dropout_rate, concat_ua=concat_ua)
                for _ in range(num_blocks)],
            autocast_dtype=None)
        n = max_sequence_len + max_output_len
        self.register_buffer("_attn_mask", torch.triu(torch.ones((n, n), dtype=torch.bool), diagonal=1))
        self._verbose = verbose
        self.reset_params()

    def reset_params(self) -> None:
        for name, params in self.named_parameters():
            if ("_hstu" in name) or ("_embedding_module" in name):
                continue
            try:
                torch.nn.init.xavier_normal_(params.data)
            except Exception:
                pass

    def get_item_embeddings(self, item_ids: torch.Tensor) -> torch.Tensor:
        return self._embedding_module.get_item_embeddings(item_ids)

    def similarity_fn(self, query_embeddings: torch.Tensor, item_ids: torch.Tensor,
                      item_embeddings: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        torch._assert(len(query_embeddings.size()) == 2, "len(query_embeddings.size()) must be 2")
        torch._assert(len(item_ids.size()) == 2, "len(item_ids.size()) must be 2")
        if item_embeddings is None:
            item_embeddings = self.get_item_embeddings(item_ids)
        torch._assert(len(item_embeddings.size()) == 3, "len(item_embeddings.size()) must be 3")
        return self._ndp_module(query_embeddings=query_embeddings, item_embeddings=item_embeddings, item_ids=item_ids, **kwargs)

    def debug_str(self) -> str:
        return (f"HSTU-b{self._num_blocks}-h{self._num_heads}-dqk{self._dqk}-dv{self._dv}"
                f"-l{self._linear_activation}d{self._linear_dropout_rate}-ad{self._attn_dropout_rate}"
                + ("" if self._enable_relative_attention_bias else "-norab"))

    def generate_user_embeddings(self, past_lengths: torch.Tensor, past_ids: torch.Tensor, past_embeddings: torch.Tensor,
                                 past_payloads: Dict[str, torch.Tensor],
                                 delta_x_offsets: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                                 cache: Optional[List[HSTUCacheState]] = None, return_cache_states: bool = False
                                 ) -> Tuple[torch.Tensor, List[HSTUCacheState]]:
        """[B, N] -> [B, N, D] (hstu.py:676-717)"""
        past_lengths, user_embeddings, _ = self._input_features_preproc(
            past_lengths=past_lengths, past_ids=past_ids, past_embeddings=past_embeddings, past_payloads=past_payloads)
        float_dtype = user_embeddings.dtype
        user_embeddings, cached_states = self._hstu(
            x=user_embeddings, x_offsets=_launch.complete_cumsum(past_lengths),
            all_timestamps=past_payloads["timestamps"] if "timestamps" in past_payloads else None,
            invalid_attn_mask=1.0 - self._attn_mask.to(float_dtype), delta_x_offsets=delta_x_offsets, cache=cache,
            return_cache_states=return_cache_states)
        return self._output_postproc(user_embeddings), cached_states

    def forward(self, past_lengths: torch.Tensor, past_ids: torch.Tensor, past_embeddings: torch.Tensor,
                past_payloads: Dict[str, torch.Tensor], batch_id: Optional[int] = None) -> torch.Tensor:
        encoded_embeddings, _ = self.generate_user_embeddings(
            past_lengths=past_lengths, past_ids=past_ids, past_embeddings=past_embeddings, past_payloads=past_payloads)
        return encoded_embeddings

    def _encode(self, past_lengths, past_ids, past_embeddings, past_payloads, delta_x_offsets, cache, return_cache_states
                ) -> Union[torch.Tensor, Tuple[torch.Tensor, List[HSTUCacheState]]]:
        encoded_seq_embeddings, cache_states = self.generate_user_embeddings(
            past_lengths=past_lengths, past_ids=past_ids, past_embeddings=past_embeddings, past_payloads=past_payloads,
            delta_x_offsets=delta_x_offsets, cache=cache, return_cache_states=return_cache_states)
        current_embeddings = get_current_embeddings(lengths=past_lengths, encoded_embeddings=encoded_seq_embeddings)
        return (current_embeddings, cache_states) if return_cache_states else current_embeddings

    def encode(self, past_lengths: torch.Tensor, past_ids: torch.Tensor, past_embeddings: torch.Tensor,
               past_payloads: Dict[str, torch.Tensor], delta_x_offsets: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
               cache: Optional[List[HSTUCacheState]] = None, return_cache_states: bool = False
               ) -> Union[torch.Tensor, Tuple[torch.Tensor, List[HSTUCacheState]]]:
        """(B, D): the encoded state at the most recent time step (hstu.py:779-809)"""
        return self._encode(past_lengths=past_lengths, past_ids=past_ids, past_embeddings=past_embeddings,
                            past_payloads=past_payloads, delta_x_offsets=delta_x_offsets, cache=cache,
                            return_cache_states=return_cache_states)
