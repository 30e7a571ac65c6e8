"""Drop-in for generative_recommenders/ops/hstu_attention.py: same function names,
argument order, defaults and assertions (``hstu_mha`` :44-128, ``delta_hstu_mha``
:131-203), backed by the HIP kernels in libhstu_hip.so.

Differences, all by construction of this package (single backend, no dispatch):
  * the ``kernel`` argument is accepted for call-site compatibility and ignored;
  * CPU tensors raise -- there is no PyTorch fallback;
  * like the reference's Triton / CUDA backends, attention dropout is not supported.
"""

from typing import Optional

import os

import torch
import torch.nn.functional as F

from generative_recommenders_amd.common import HammerKernel
from generative_recommenders_amd.ops import _launch


def _pad_head_dim(t: torch.Tensor) -> torch.Tensor:
    """Head dims that are not a multiple of the 16-byte vector (e.g. 50, 25 in the shipped
    ML-1M configs) are zero-padded; zeros change neither q.k nor the sliced output."""
    mult = 16 // t.element_size()
    pad = (-t.shape[-1]) % mult
    return F.pad(t, (0, pad)) if pad else t


_PRECISE = os.environ.get("HSTU_ATTN_PRECISE") == "1"      # read once, as the library does (csrc/attn_misc.hip, attn_fwd_precise_enabled)


class _HstuMhaFunction(torch.autograd.Function):
    """Autograd node of the fused attention: saves q, k, v and the jagged metadata and
    recomputes S in the backward kernel (same contract as _AttentionFunction,
    ops/triton/triton_hstu_attention.py:1951-2063)."""

    @staticmethod
    def forward(ctx, max_seq_len, alpha, q, k, v, seq_offsets, num_targets, max_attn_len, contextual_seq_len,
                min_full_attn_seq_len, user_order=None):
        out = _launch.attn_fwd(q, k, v, seq_offsets, num_targets, max_seq_len, alpha, 1.0 / max_seq_len,
                               max_attn_len, contextual_seq_len, min_full_attn_seq_len, user_order=user_order)
        saved = [q, k, v, seq_offsets] + ([num_targets] if num_targets is not None else [])
        ctx.save_for_backward(*saved)
        ctx.user_order = user_order
        ctx.has_targets = num_targets is not None
        ctx.args = (max_seq_len, alpha, max_attn_len, contextual_seq_len, min_full_attn_seq_len)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, seq_offsets = ctx.saved_tensors[:4]
        num_targets = ctx.saved_tensors[4] if ctx.has_targets else None
        max_seq_len, alpha, max_attn_len, contextual_seq_len, min_full = ctx.args
        if _PRECISE and q.dtype in (torch.bfloat16, torch.float16):
            # HSTU_ATTN_PRECISE=1, backward: the 16-bit kernels round P' and dS' to the I/O dtype in front of their second MFMAs (one
            # rounding of the size of the output's own; the folded kernel has no registers for a second fragment pair: 256 of 256).
            # The precise backward is the fp32 instantiation of the same kernels on the same (exactly representable) inputs, its
            # gradients rounded ONCE to the I/O dtype: error = the output rounding alone.  Several times the time and four fp32 copies;
            # a numerical mode, not the measured path.
            dq, dk, dv = _launch.attn_bwd(dout.float(), q.float(), k.float(), v.float(), seq_offsets, num_targets, max_seq_len, alpha,
                                          1.0 / max_seq_len, max_attn_len, contextual_seq_len, min_full, user_order=ctx.user_order)
            return None, None, dq.to(q.dtype), dk.to(q.dtype), dv.to(q.dtype), None, None, None, None, None, None
        dq, dk, dv = _launch.attn_bwd(dout, q, k, v, seq_offsets, num_targets, max_seq_len, alpha,
                                      1.0 / max_seq_len, max_attn_len, contextual_seq_len, min_full, user_order=ctx.user_order)
        return None, None, dq, dk, dv, None, None, None, None, None, None


def hip_hstu_mha(max_seq_len, alpha, q, k, v, seq_offsets, num_targets=None, max_attn_len=0,
                 contextual_seq_len=0, min_full_attn_seq_len=0, sort_by_length=False) -> torch.Tensor:
    dqk, dv = q.shape[2], v.shape[2]
    qp, kp, vp = _pad_head_dim(q), _pad_head_dim(k), _pad_head_dim(v)
    # sort_by_length: workgroups take the users in descending-length order (heavy first), as the reference's Triton
    # launch does (triton_hstu_attention.py:1968-1973); results do not depend on it
    order = _launch.length_order(_launch._idx(seq_offsets)) if sort_by_length and seq_offsets.numel() > 2 else None
    out = _HstuMhaFunction.apply(max_seq_len, alpha, qp, kp, vp, seq_offsets, num_targets, max_attn_len,
                                 contextual_seq_len, min_full_attn_seq_len, order)
    del dqk
    return out[..., :dv] if out.shape[2] != dv else out


def hstu_mha(
    max_seq_len: int,
    alpha: float,
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    seq_offsets: torch.Tensor,
    causal: bool = True,
    dropout_pr: float = 0.0,
    training: bool = True,
    num_targets: Optional[torch.Tensor] = None,
    max_attn_len: int = 0,
    contextual_seq_len: int = 0,
    min_full_attn_seq_len: int = 0,
    sort_by_length: bool = False,
    kernel: HammerKernel = HammerKernel.HIP,
    enable_tma: bool = False,
) -> torch.Tensor:
    _, H, _ = q.shape
    torch._assert(max_seq_len > 0, "max_seq_len must be larger than 0")
    torch._assert(q.dim() == 3, "q must be 3-D")
    torch._assert(k.shape == q.shape, "k must be the same shape as q")
    torch._assert(v.dim() == 3, "v must be 3-D")
    torch._assert(v.shape[0] == q.shape[0], "wrong v shape[0]")
    torch._assert(v.shape[1] == H, "wrong v shape[1]")
    torch._assert(causal, "only support causal attention")
    torch._assert(dropout_pr < 1e-6, "dropout for the HIP path not implemented")
    torch._assert(max_attn_len >= 0 and contextual_seq_len >= 0 and min_full_attn_seq_len >= 0,
                  "mask parameters must be non-negative")
    # enable_tma has no gfx950 meaning
    del training, kernel, enable_tma
    return hip_hstu_mha(max_seq_len, alpha, q, k, v, seq_offsets, num_targets, max_attn_len,
                        contextual_seq_len, min_full_attn_seq_len, sort_by_length)


def delta_hstu_mha(
    max_seq_len: int,
    alpha: float,
    delta_q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    seq_offsets: torch.Tensor,
    num_targets: Optional[torch.Tensor] = None,
    max_attn_len: int = 0,
    contextual_seq_len: int = 0,
    kernel: HammerKernel = HammerKernel.HIP,
    enable_tma: bool = False,
) -> torch.Tensor:
    L, H, D = delta_q.shape
    B = seq_offsets.size(0) - 1
    torch._assert(max_seq_len > 0, "max_seq_len must be larger than 0")
    torch._assert(delta_q.dim() == 3, "delta_q must be 3-D")
    torch._assert(B > 0 and L % B == 0, "delta_q must be padded")
    torch._assert(k.dim() == 3, "k must be 3-D")
    torch._assert(k.shape[1] == H, "wrong k shape[1]")
    torch._assert(k.shape[2] == D, "wrong k shape[2]")
    torch._assert(v.dim() == 3, "v must be 3-D")
    torch._assert(v.shape[1] == H, "wrong v shape[1]")
    del kernel, enable_tma
    dv = v.shape[2]
    out = _launch.attn_fwd(_pad_head_dim(delta_q), _pad_head_dim(k), _pad_head_dim(v), seq_offsets, num_targets,
                           max_seq_len, alpha, 1.0 / max_seq_len, max_attn_len, contextual_seq_len, 0,
                           delta_q=L // B)
    return out[..., :dv] if out.shape[2] != dv else out
