"""Drop-in for the research-path HSTU layer of
generative_recommenders/research/modeling/sequential/hstu.py: the relative attention bias
modules (:51-144), the attention (:150-223) and ``SequentialTransductionUnitJagged`` (:226-444),
on the fused HIP kernels: the (B, N, N) bias and the (B, H, N, N) logits of the reference are
never materialised -- the bias is generated inside the attention kernels from
``(pos_w, ts_w, timestamps)`` and its gradient is reduced per workgroup.

Parameter names match the reference (``_uvqk``, ``_o.weight``, ``_o.bias``,
``_rel_attn_bias._ts_w`` / ``_pos_w`` / ``_w``) so its state_dicts load unchanged.
Not supported (the reference has no caller for them either): the
``delta_x_offsets`` / ``cache`` incremental branch and the ``softmax_rel_bias`` normalisation.
"""

import abc
import ctypes as C
import math
from typing import Callable, Optional, Tuple

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


class _RelBiasAttentionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, q, k, v, x_offsets, timestamps, pos_w, ts_w, num_buckets, bucket_div):
        for name, t in (("q", q), ("k", k), ("v", v), ("x_offsets", x_offsets), ("pos_w", pos_w)):
            L.require_gpu_tensor(t, name)
        q, k, v = _launch._aligned_rows(q), _launch._aligned_rows(k), _launch._aligned_rows(v)
        x_offsets = _launch._idx(x_offsets)
        pos32 = pos_w.detach().float().contiguous()
        ts32 = None if ts_w is None else ts_w.detach().float().contiguous()
        ts = None if ts_w is None else timestamps.to(torch.int64).contiguous()
        torch._assert(pos32.numel() == 2 * n - 1, "pos_w must have 2 * n - 1 entries")
        out = torch.empty((q.shape[0], q.shape[1], v.shape[2]), dtype=q.dtype, device=q.device)
        if q.shape[0]:
            p = L.HstuAttnParams()
            _launch._fill_attn_params(p, q, k, v, out, x_offsets, None, n, 1.0, 1.0 / n, 0, 0, 0, 0)
            _fill_bias(p, pos32, ts32, ts, num_buckets, bucket_div)
            with torch.cuda.device(q.device):
                L.check(L.lib().hstu_attn_fwd(C.byref(p), L.current_stream_ptr(q.device)))
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
            _launch._fill_attn_params(bp.fwd, q, k, v, None, x_offsets, None, n, 1.0, 1.0 / n, 0, 0, 0, 0)
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


def hstu_rel_bias_attention(num_heads: int, attention_dim: int, linear_dim: int, q: torch.Tensor, k: torch.Tensor,
                            v: torch.Tensor, x_offsets: torch.Tensor, all_timestamps: Optional[torch.Tensor], n: int,
                            rel_attn_bias: RelativeAttentionBiasModule) -> torch.Tensor:
    """Fused equivalent of ``_hstu_attention_maybe_from_cache`` (hstu.py:150-223) for the full-sequence
    case: q, k (sum L, H*attention_dim), v (sum L, H*linear_dim) -> (sum L, H*linear_dim);
    P = silu(q.k + rel_bias) / n with the lower-triangular (diagonal included) mask."""
    pos_w, ts_w, nb, div = rel_attn_bias.bias_params()
    if all_timestamps is None:
        ts_w = None
    L_ = q.shape[0]
    q3 = _pad_head_dim(q.view(L_, num_heads, attention_dim))
    k3 = _pad_head_dim(k.view(L_, num_heads, attention_dim))
    v3 = _pad_head_dim(v.view(L_, num_heads, linear_dim))
    dpad = max(q3.shape[2], v3.shape[2])
    # the bias kernels are instantiated for dqk == dv: pad the smaller head dim with zeros
    if q3.shape[2] != dpad:
        q3, k3 = F.pad(q3, (0, dpad - q3.shape[2])), F.pad(k3, (0, dpad - k3.shape[2]))
    if v3.shape[2] != dpad:
        v3 = F.pad(v3, (0, dpad - v3.shape[2]))
    out = _RelBiasAttentionFunction.apply(n, q3, k3, v3, x_offsets, all_timestamps, pos_w, ts_w, nb, div)
    return out[..., :linear_dim].reshape(L_, num_heads * linear_dim)


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

    def forward(self, x: torch.Tensor, x_offsets: torch.Tensor, all_timestamps: Optional[torch.Tensor],
                invalid_attn_mask: torch.Tensor, delta_x_offsets=None, cache=None, return_cache_states: bool = False):
        """x (sum L, D) -> x' (sum L, D); ``invalid_attn_mask`` only supplies n (the kernels apply the
        lower-triangular mask the reference registers, hstu.py:626-638)."""
        from generative_recommenders_amd.ops.hstu_compute import _NormMulFunction, _SiluFunction
        from generative_recommenders_amd.ops.layer_norm import layer_norm

        if delta_x_offsets is not None or cache is not None:
            raise NotImplementedError("incremental (delta_x_offsets / cache) decoding is not supported")
        assert self._rel_attn_bias is not None
        n = invalid_attn_mask.size(-1)
        D, H, Ld, A = self._embedding_dim, self._num_heads, self._linear_dim, self._attention_dim
        ones = torch.ones(D, dtype=x.dtype, device=x.device)
        normed_x = layer_norm(x, ones, torch.zeros_like(ones), self._eps)          # LN without affine
        mm = torch.mm(normed_x, self._uvqk.to(x.dtype))
        if self._linear_activation == "silu":
            mm = _SiluFunction.apply(mm)                                            # SiLU on all of u, v, q, k
        elif self._linear_activation != "none":
            raise ValueError(f"Unknown linear_activation {self._linear_activation}")
        u, v, q, k = torch.split(mm, [Ld * H, Ld * H, A * H, A * H], dim=1)
        attn = hstu_rel_bias_attention(H, A, Ld, q, k, v, x_offsets, all_timestamps, n, self._rel_attn_bias)
        w1 = torch.ones(Ld * H, dtype=x.dtype, device=x.device)
        if self._concat_ua:
            a = layer_norm(attn, w1, torch.zeros_like(w1), self._eps)
            o_input = torch.cat([u, a, u * a], dim=-1)
        else:
            o_input = _NormMulFunction.apply(attn, u.contiguous(), w1, torch.zeros_like(w1), self._eps, H, Ld, False,
                                             False)
        new_outputs = self._o(F.dropout(o_input, p=self._dropout_ratio, training=self.training)) + x
        return new_outputs, (v, None, None, new_outputs)
