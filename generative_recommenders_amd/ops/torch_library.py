"""Operator seam: registers the reference's ``hstu::`` torch.library schemas on top of the HIP
library, so code that calls ``torch.ops.hstu.*`` (cuda_hstu_attention.py:50,
cuda_hstu_preprocess_and_attention.py:114,280, ops/benchmarks/hstu_attention_bench.py:263,290)
runs unchanged on MI355X.

Schemas are the reference's, argument for argument:
  hstu_mha / hstu_mha_fwd / hstu_mha_bwd      ops/cpp/hstu_attention/flash_api.cpp:275-352
  complete_cumsum, expand_1d_jagged_to_dense, concat_1d_jagged_jagged     ops/cpp/cpp_ops.cpp:94-102
Dispatch keys: CUDA (== HIP tensors under PyTorch-ROCm) -> libhstu_hip.so; Meta -> shape functions
(the reference ships those for tracing, flash_meta.cpp); CPU is deliberately NOT registered (the
reference's CPU entry is a dummy returning empty tensors, flash_cpu_dummy.cpp; ours fails loudly).
fp8 descale / attn_scale tensors, which no Python caller in the reference passes, are rejected.
"""

from typing import List, Optional, Tuple

import torch

from generative_recommenders_amd.ops import _launch
from generative_recommenders_amd.ops.hstu_attention import _pad_head_dim

_REGISTERED = False
_lib_def = None
_keep = []


def _check_unsupported(attn_scale, q_descale=None, k_descale=None, v_descale=None):
    if attn_scale is not None:
        raise RuntimeError("hstu_mha: per-row attn_scale is not supported by the HIP backend")
    if q_descale is not None or k_descale is not None or v_descale is not None:
        raise RuntimeError("hstu_mha: fp8 descale tensors are not supported by the HIP backend")


def _fwd_impl(max_seq_len, alpha, q, k, v, seq_offsets, causal, num_targets, attn_scale, max_attn_len,
              min_full_attn_seq_len, contextual_seq_len, q_descale, k_descale, v_descale, sm_margin):
    _check_unsupported(attn_scale, q_descale, k_descale, v_descale)
    torch._assert(causal, "only support causal attention")
    torch._assert(seq_offsets is not None, "the HIP backend takes jagged (seq_offsets) inputs")
    dv = v.shape[2]
    out = _launch.attn_fwd(_pad_head_dim(q), _pad_head_dim(k), _pad_head_dim(v), seq_offsets, num_targets,
                           int(max_seq_len), alpha, 1.0 / int(max_seq_len), max_attn_len, contextual_seq_len,
                           min_full_attn_seq_len)
    return out[..., :dv].contiguous() if out.shape[2] != dv else out


def _bwd_impl(max_seq_len, alpha, dout, q, k, v, dq, dk, dv, seq_offsets, causal, num_targets, attn_scale,
              max_attn_len, min_full_attn_seq_len, contextual_seq_len, sort_by_length, deterministic, sm_margin):
    _check_unsupported(attn_scale)
    torch._assert(causal, "only support causal attention")
    es = q.element_size()
    if (q.shape[2] * es) % 16 or (v.shape[2] * es) % 16:
        gq, gk, gv = _launch.attn_bwd(_pad_head_dim(dout), _pad_head_dim(q), _pad_head_dim(k), _pad_head_dim(v),
                                      seq_offsets, num_targets, max_seq_len, alpha, 1.0 / max_seq_len, max_attn_len,
                                      contextual_seq_len, min_full_attn_seq_len)
        dq.copy_(gq[..., : q.shape[2]]); dk.copy_(gk[..., : k.shape[2]]); dv.copy_(gv[..., : v.shape[2]])
    else:  # write straight into the caller's (possibly strided) dq / dk / dv, as the reference does
        _launch.attn_bwd(dout, q, k, v, seq_offsets, num_targets, max_seq_len, alpha, 1.0 / max_seq_len, max_attn_len,
                         contextual_seq_len, min_full_attn_seq_len, dq=dq, dk=dk, dv=dv)
    return [dq, dk, dv]


class _HstuMhaOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, max_seq_len, alpha, q, k, v, seq_offsets, causal, num_targets, attn_scale, max_attn_len,
                min_full_attn_seq_len, contextual_seq_len, q_descale, k_descale, v_descale, sort_by_length,
                deterministic, sm_margin):
        out = torch.ops.hstu.hstu_mha_fwd(max_seq_len, alpha, q, k, v, seq_offsets, causal, num_targets, attn_scale,
                                          max_attn_len, min_full_attn_seq_len, contextual_seq_len, q_descale, k_descale,
                                          v_descale, sm_margin)
        ctx.save_for_backward(q, k, v, seq_offsets, *([num_targets] if num_targets is not None else []))
        ctx.has_targets = num_targets is not None
        ctx.meta = (int(max_seq_len), alpha, causal, max_attn_len, min_full_attn_seq_len, contextual_seq_len,
                    sort_by_length, deterministic, sm_margin)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, seq_offsets = ctx.saved_tensors[:4]
        nt = ctx.saved_tensors[4] if ctx.has_targets else None
        N, alpha, causal, w, f, c, sbl, det, smm = ctx.meta
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        torch.ops.hstu.hstu_mha_bwd(N, alpha, dout.contiguous(), q, k, v, dq, dk, dv, seq_offsets, causal, nt, None, w, f,
                                    c, sbl, det, smm)
        return (None, None, dq, dk, dv) + (None,) * 13


def _mha_impl(*args):
    return _HstuMhaOp.apply(*args)


def sort_kv_pairs(keys: torch.Tensor, values: torch.Tensor, end_bit=None, descending: bool = False):
    """``hstu::sort_kv_pairs`` (ops/cpp/cpp_ops.cpp:73-101, sort_kv_pairs_cuda_kernels_template.cu:9-77): stable
    radix sort of 1-D (key, value) pairs on key bits [0, end_bit) -- all bits when ``end_bit`` is None.  Index
    plumbing: the sort itself is torch's (rocPRIM radix sort on the GPU), as the reference's is cub's."""
    if keys.dtype not in (torch.int32, torch.int64, torch.uint8, torch.int16):
        raise RuntimeError("sort_kv_pairs: keys must be int32, int64, uint8 or int16")
    if keys.dim() != 1 or values.dim() != 1 or keys.shape != values.shape:
        raise RuntimeError("sort_kv_pairs: keys and values must be 1-D tensors of one length")
    width = keys.element_size() * 8
    if end_bit is None or end_bit >= width:
        sub = keys
    else:
        if end_bit <= 0:
            return keys.clone(), values.clone()          # no key bits: a stable sort leaves the order alone
        sub = keys.to(torch.int64) & ((1 << int(end_bit)) - 1)
    order = torch.sort(sub, stable=True, descending=bool(descending)).indices
    return keys[order], values[order]


def register() -> None:
    """Idempotent; raises if another library (e.g. the reference's CUDA extension) already owns
    the ``hstu`` schemas."""
    global _REGISTERED, _lib_def
    if _REGISTERED:
        return
    lib = torch.library.Library("hstu", "FRAGMENT")
    _lib_def = lib
    lib.define(
        "hstu_mha(SymInt max_seq_len, float alpha, Tensor q, Tensor k, Tensor v, Tensor? seq_offsets, bool causal, "
        "Tensor? num_targets, Tensor? attn_scale, int max_attn_len, int min_full_attn_seq_len, "
        "int contextual_seq_len, Tensor? q_descale, Tensor? k_descale, Tensor? v_descale, bool sort_by_length, "
        "bool deterministic, int sm_margin) -> Tensor")
    lib.define(
        "hstu_mha_fwd(SymInt max_seq_len, float alpha, Tensor q, Tensor k, Tensor v, Tensor? seq_offsets, bool causal, "
        "Tensor? num_targets, Tensor? attn_scale, int max_attn_len, int min_full_attn_seq_len, "
        "int contextual_seq_len, Tensor? q_descale, Tensor? k_descale, Tensor? v_descale, int sm_margin) -> Tensor")
    lib.define(
        "hstu_mha_bwd(int max_seq_len, float alpha, Tensor dout, Tensor q, Tensor k, Tensor v, Tensor(a!) dq, "
        "Tensor(b!) dk, Tensor(c!) dv, Tensor? seq_offsets, bool causal, Tensor? num_targets, Tensor? attn_scale, "
        "int max_attn_len, int min_full_attn_seq_len, int contextual_seq_len, bool sort_by_length,"
        "bool deterministic,int sm_margin) -> Tensor[]")
    lib.define("complete_cumsum(Tensor values) -> Tensor")
    lib.define("expand_1d_jagged_to_dense(Tensor values, Tensor offsets, SymInt max_len) -> Tensor")
    lib.define("concat_1d_jagged_jagged(Tensor lengths_left, Tensor values_left, Tensor lengths_right, "
               "Tensor values_right) -> Tensor")

    lib.define("sort_kv_pairs(Tensor keys, Tensor values, int? end_bit=None, bool descending=False) -> (Tensor, Tensor)")

    lib.impl("hstu_mha_fwd", _fwd_impl, "CUDA")
    lib.impl("hstu_mha_bwd", _bwd_impl, "CUDA")
    lib.impl("hstu_mha", _mha_impl, "CompositeImplicitAutograd")
    lib.impl("complete_cumsum", lambda values: _launch.complete_cumsum(values), "CUDA")
    lib.impl("expand_1d_jagged_to_dense",
             lambda values, offsets, max_len: _launch.expand_1d_jagged_to_dense(values, offsets, int(max_len)), "CUDA")
    lib.impl("concat_1d_jagged_jagged",
             lambda ll, vl, lr, vr: _launch.concat_1d_jagged_jagged(ll, vl, lr, vr), "CUDA")

    lib.impl("sort_kv_pairs", sort_kv_pairs, "CompositeExplicitAutograd")

    # Meta (shape-only) kernels, as flash_meta.cpp provides for tracing
    def _fwd_meta(max_seq_len, alpha, q, k, v, *rest):
        return q.new_empty((q.shape[0], q.shape[1], v.shape[2]))

    lib.impl("hstu_mha_fwd", _fwd_meta, "Meta")
    lib.impl("complete_cumsum", lambda values: values.new_empty((values.shape[0] + 1,)), "Meta")
    lib.impl("expand_1d_jagged_to_dense",
             lambda values, offsets, max_len: values.new_empty((offsets.shape[0] - 1, max_len)), "Meta")
    _REGISTERED = True
