"""Drop-in for generative_recommenders/ops/hstu_compute.py: ``hstu_compute_uqvk`` (:50-89),
``hstu_compute_output`` (:92-168), ``hstu_preprocess_and_attention`` (:171-259).

The two GEMMs go to hipBLASLt via torch.addmm / torch.mm (MFMA; what the reference does on
AMD); everything around them -- LayerNorm, SiLU(u), the jagged attention, LN/GroupNorm * u
with the [u, attn, y] concat -- runs on the HIP kernels of libhstu_hip.so.  The two fused
autograd nodes follow the saved-tensor / recompute contract of the reference's Triton path
(triton_hstu_preprocess_and_attention.py:37-293, triton_hstu_linear.py:1137-1308;
SURVEY.md App. F): results are identical whichever recompute flags are chosen.
"""

import os
import weakref
from typing import Optional, Tuple

import torch

from generative_recommenders_amd.common import HammerKernel
from generative_recommenders_amd.ops import _launch
from generative_recommenders_amd.ops.hstu_attention import hstu_mha
from generative_recommenders_amd.ops.layer_norm import layer_norm
from generative_recommenders_amd.ops.mm import weight_grad_mm


def hstu_compute_uqvk(
    x: torch.Tensor,
    norm_weight: torch.Tensor,
    norm_bias: torch.Tensor,
    norm_eps: float,
    num_heads: int,
    attn_dim: int,
    hidden_dim: int,
    uvqk_weight: torch.Tensor,
    uvqk_bias: torch.Tensor,
    kernel: HammerKernel = HammerKernel.HIP,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """LN_affine(x) @ W + b, split as [u, v, q, k], SiLU on u only."""
    del kernel
    if _LN_LINEAR and _UVQK_LINEAR and x.is_cuda and x.dim() == 2 and _launch.ln_linear_supported(x, uvqk_weight.shape[1]):
        # the fused LayerNorm + projection kernel (one autograd node), as the STU layer's nodes use it: the K / V rows a delta
        # call appends to a cache are then bit-identical to the ones the prefill wrote (a row's result does not depend on its
        # neighbours in the batch)
        uvqk = _LnUvqkFunction.apply(x, norm_weight, norm_bias, uvqk_weight, uvqk_bias, norm_eps, torch.is_grad_enabled())
    else:
        norm_weight, norm_bias, uvqk_weight, uvqk_bias = (t.to(x.dtype) for t in (norm_weight, norm_bias, uvqk_weight, uvqk_bias))
        normed_x = layer_norm(x, weight=norm_weight, bias=norm_bias, eps=norm_eps)
        # (autograd's own nodes.  The same GEMM kernel as the fused layer's -- linear() on a K-contiguous weight -- so that the
        # K / V rows a delta call appends to a cache are bit-identical to the ones the prefill wrote)
        if _UVQK_LINEAR and normed_x.is_cuda:
            tracked = torch.is_grad_enabled() and uvqk_weight.requires_grad
            wt = uvqk_weight.t().contiguous() if tracked else _kmajor(uvqk_weight, normed_x.dtype)
            uvqk = torch.nn.functional.linear(normed_x, wt.to(normed_x.dtype), uvqk_bias.to(normed_x.dtype))
        else:
            uvqk = torch.addmm(uvqk_bias, normed_x, uvqk_weight)
    u, v, q, k = torch.split(
        uvqk, [hidden_dim * num_heads, hidden_dim * num_heads, attn_dim * num_heads, attn_dim * num_heads], dim=1
    )
    u = _SiluFunction.apply(u)
    q = q.view(-1, num_heads, attn_dim)
    k = k.view(-1, num_heads, attn_dim)
    v = v.view(-1, num_heads, hidden_dim)
    return u, q, k, v


class _LnUvqkFunction(torch.autograd.Function):
    """uvqk = LayerNorm(x) @ W + b by the fused kernel (csrc/hstu_ln_linear.cuh); backward = the projection's three GEMM-shaped
    gradients (hipBLASLt) + the layer-norm backward kernel, normed_x recomputed by the row kernel (as the STU layer's nodes do)."""

    @staticmethod
    def forward(ctx, x, norm_weight, norm_bias, uvqk_weight, uvqk_bias, eps, grad_on=True):
        # (``grad_on``: the caller's grad mode -- inside forward it is always off, and needs_input_grad ignores it)
        ctx.param_dtypes = (norm_weight.dtype, norm_bias.dtype, uvqk_weight.dtype, uvqk_bias.dtype)
        (nw, nb, w, beta), kmajor = _prepare_params((norm_weight, norm_bias, uvqk_weight, uvqk_bias), x.dtype, 2,
                                                    grad_on and any(ctx.needs_input_grad))
        uvqk, _, mean, rstd = _ln_uvqk(x, nw, nb, eps, w, kmajor, beta, want_normed=False)
        ctx.save_for_backward(x, nw, nb, mean, rstd, w)
        ctx.kmajor, ctx.eps = kmajor, eps
        return uvqk

    @staticmethod
    def backward(ctx, duvqk):
        x, nw, nb, mean, rstd, w = ctx.saved_tensors
        duvqk = duvqk.contiguous()
        normed_x, _, _ = _launch.layer_norm_fwd(x, nw, nb, ctx.eps)
        dt = ctx.param_dtypes
        dbeta = _bias_grad(duvqk, dt[3])
        d_normed = _uvqk_dgrad(duvqk, w, ctx.kmajor)
        dW = weight_grad_mm(normed_x, duvqk, out_dtype=dt[2])
        dx, dnw, dnb = _launch.layer_norm_bwd(d_normed, x, nw, mean, rstd)
        return dx, dnw.to(dt[0]), dnb.to(dt[1]), dW, dbeta.to(dt[3]), None, None


class _SiluFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return _launch.silu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _launch.silu_bwd(dy, x)


def draw_dropout_seed() -> int:
    """The seed of one fused-dropout call, drawn the way the reference draws it (triton_hstu_linear.py:376-377: torch's
    default CPU generator, no device sync) -- ``torch.manual_seed`` makes a run reproducible."""
    return int(torch.randint(low=0, high=2**62, size=(1,), dtype=torch.int64).item())


# The UVQK projection as ``linear(x, W^T-copy, b)``: hipBLASLt's kernels for a K-contiguous weight run this shape (K = 512, N =
# 2048) at 797 TFLOP/s against 737 for ``addmm(b, x, W)`` with the reference's (in, out) layout (tools/bench_gemm_layout.py,
# profiles/r03_gemm_layout.txt).  The transposed copy IS the parameter cast (one pass, cached per parameter version: _kmajor) and
# it is what backward keeps: the data gradient is mm(duvqk, W^T-copy) without another transpose.  HSTU_UVQK_LINEAR=0 keeps addmm.
_UVQK_LINEAR = os.environ.get("HSTU_UVQK_LINEAR", "1") != "0"


_KMAJOR_CACHE: dict = {}      # id(parameter) -> (weak reference, (version, dtype, data_ptr), copy); tensors compare element-wise,
                               # so they cannot key a (weak) dictionary themselves
_PARAM_CACHE: dict = {}       # ids of a layer's parameters -> (weak references, key, casted copies): inference only
_PARAM_CACHE_ON = os.environ.get("HSTU_PARAM_CACHE", "1") != "0"      # (inference-time cache of low-precision parameter copies)


def invalidate_parameter_caches() -> None:
    """Drop every cached low-precision copy of a parameter.  The caches serve calls that do NOT track gradients (inference:
    24 layers x 7 parameters would otherwise be re-cast for every microbatch) and are keyed on the parameters' version
    counters, which an optimizer step, ``load_state_dict`` or any in-place op on the parameter bumps.  Writing through
    ``parameter.data`` (``p.data.mul_(..)``, weight EMA swaps by ``.data`` assignment of the same storage) does NOT bump
    the counter: call this afterwards.  Calls that track gradients (training) never read the caches."""
    _KMAJOR_CACHE.clear()
    _PARAM_CACHE.clear()


def _kmajor(weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """The (in, out) projection weight as a K-contiguous (out, in) copy in the activations' dtype: ONE transposing, casting
    pass, cached per parameter until the parameter changes (its version counter: an optimizer step, a load_state_dict).
    Only calls that do not track gradients come here (see invalidate_parameter_caches for the ``.data`` caveat)."""
    wid = id(weight)
    key = (weight._version, dtype, weight.data_ptr())
    hit = _KMAJOR_CACHE.get(wid)
    if hit is not None and hit[0]() is weight and hit[1] == key:
        return hit[2]
    wt = torch.empty((weight.shape[1], weight.shape[0]), dtype=dtype, device=weight.device)
    wt.copy_(weight.detach().t())
    _KMAJOR_CACHE[wid] = (weakref.ref(weight, lambda _r, wid=wid: _KMAJOR_CACHE.pop(wid, None)), key, wt)
    return wt


def _cast_fresh(params, dtype, kmajor_index):
    """the parameters in ``dtype``, cast NOW: every fp32 CUDA parameter in one launch (hstu_cast_params; the one at
    ``kmajor_index`` as its K-contiguous transpose), anything else by torch"""
    outs = [None] * len(params)
    batch = [i for i, t in enumerate(params)
             if t.is_cuda and t.dtype == torch.float32 and dtype in (torch.bfloat16, torch.float16) and t.numel() > 0]
    if len(batch) >= 2 or (kmajor_index is not None and kmajor_index in batch):
        res = _launch.cast_params([params[i] for i in batch], dtype,
                                  transpose_index=batch.index(kmajor_index) if kmajor_index in batch else None)
        for i, r in zip(batch, res):
            outs[i] = r
    for i, t in enumerate(params):
        if outs[i] is None:
            if i == kmajor_index:
                outs[i] = torch.empty((t.shape[1], t.shape[0]), dtype=dtype, device=t.device)
                outs[i].copy_(t.detach().t())
            else:
                outs[i] = _cast(t, dtype)
    return outs


def _prepare_params(params, dtype, kmajor_index, tracked):
    """(copies, kmajor): a node's parameters in the activations' dtype, the UVQK weight (``kmajor_index``) as its K-contiguous
    copy where hipBLASLt / the fused kernel want it.  ``tracked`` (the call records a graph -- ANY input wants a gradient, the
    activations included: training, fine-tuning with frozen layers): cast now, every time -- a cache keyed on the version counter
    would miss writes through ``.data`` (legacy optimizers, EMA swaps) and multiply by a stale weight silently; one launch per node
    covers all of them.  Otherwise (inference) the copies are cached per parameter set until a version counter moves
    (invalidate_parameter_caches; ``HSTU_PARAM_CACHE=0`` turns the cache off)."""
    want_kmajor = kmajor_index is not None and _UVQK_LINEAR and params[kmajor_index].is_cuda
    kidx = kmajor_index if want_kmajor else None
    if tracked or not _PARAM_CACHE_ON:
        return _cast_fresh(params, dtype, kidx), want_kmajor
    ids = tuple(id(t) for t in params)
    try:
        key = tuple((t._version, t.data_ptr(), t.dtype) for t in params) + (dtype, want_kmajor)
    except RuntimeError:        # (a parameter created under torch.inference_mode has no version counter)
        return _cast_fresh(params, dtype, kidx), want_kmajor
    hit = _PARAM_CACHE.get(ids)
    if hit is not None and hit[1] == key and all(r() is t for r, t in zip(hit[0], params)):
        return hit[2], want_kmajor
    outs = _cast_fresh(params, dtype, kidx)
    refs = tuple(weakref.ref(t, lambda _r, ids=ids: _PARAM_CACHE.pop(ids, None)) for t in params)
    _PARAM_CACHE[ids] = (refs, key, outs)
    return outs, want_kmajor


# d uvqk_beta = the column sums of d uvqk (triton_addmm.py:309): one HBM-bound read of (rows, 2048) 16-bit values
# (hstu_column_sum, fixed summation order: 152-157 us at 204,800 x 2048 = 5.1-5.2 TB/s).  (Launched on a side stream under the two
# MFMA-bound GEMMs that read d uvqk next it measured no gain -- layer step 13.86 vs 13.88 ms, docs/EXPERIMENTS.md R5.4 -- so it
# runs in stream order and the side-stream variant is gone.)
def _bias_grad(duvqk: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    """fp32 column sums of d uvqk (torch's reduction for layouts / dtypes the kernel does not take)"""
    if _launch.column_sum_supported(duvqk):
        return _launch.column_sum(duvqk)
    return duvqk.sum(dim=0, dtype=torch.float32 if out_dtype == torch.float32 else None)


def _uvqk_prepare(weight: torch.Tensor, dtype: torch.dtype):
    """(tensor, kmajor): what the fused nodes multiply by and save for backward"""
    if _UVQK_LINEAR and weight.is_cuda:
        return _kmajor(weight, dtype), True
    return _cast(weight, dtype), False


def _uvqk_gemm(normed_x: torch.Tensor, w: torch.Tensor, kmajor: bool, bias: torch.Tensor) -> torch.Tensor:
    return torch.nn.functional.linear(normed_x, w, bias) if kmajor else torch.addmm(bias, normed_x, w)


# LayerNorm + the UVQK GEMM as ONE hand-written kernel (csrc/hstu_ln_linear.cuh: x read once, normalised in registers, the
# weight streamed through LDS) where its shape conditions hold (16-bit activations, embedding dim 512, K-major weight):
# 442 us against 82 + 577 us for hstu_layer_norm_fwd + hipBLASLt at 204,800 rows x 2048 columns
# (profiles/r04_ln_linear_bench.txt).  HSTU_LN_LINEAR=0 keeps the two calls.
_LN_LINEAR = os.environ.get("HSTU_LN_LINEAR", "1") != "0"


def _ln_uvqk(x, norm_weight, norm_bias, eps, w, kmajor, bias, want_normed):
    """(uvqk, normed_x or None, mean, rstd): the fused kernel when it takes the shape, else layer norm + GEMM"""
    if _LN_LINEAR and kmajor and _launch.ln_linear_supported(x, w.shape[0]):
        return _launch.ln_linear_fwd(x, norm_weight, norm_bias, eps, w, bias, want_normed=want_normed)
    normed_x, mean, rstd = _launch.layer_norm_fwd(x, norm_weight, norm_bias, eps)
    return _uvqk_gemm(normed_x, w, kmajor, bias), normed_x, mean, rstd


def _recompute_uvqk(x, norm_weight, norm_bias, eps, w, kmajor, bias, normed_x):
    """backward's (uvqk, normed_x) when uvqk was not kept: by the SAME kernel as forward -- the fused kernel and hipBLASLt sum
    in different orders, and a recomputed u that differs from forward's in the last bit makes the one-node and two-node layers
    disagree -- so where forward took the fused kernel it runs again here even if normed_x was kept (it is the faster call anyway)"""
    if _LN_LINEAR and kmajor and _launch.ln_linear_supported(x, w.shape[0]):
        uvqk, nx, _, _ = _launch.ln_linear_fwd(x, norm_weight, norm_bias, eps, w, bias, want_normed=normed_x is None)
        return uvqk, (nx if normed_x is None else normed_x)
    if normed_x is None:
        normed_x, _, _ = _launch.layer_norm_fwd(x, norm_weight, norm_bias, eps)
    return _uvqk_gemm(normed_x, w, kmajor, bias), normed_x


def _uvqk_dgrad(duvqk: torch.Tensor, w: torch.Tensor, kmajor: bool) -> torch.Tensor:
    return torch.mm(duvqk, w) if kmajor else torch.mm(duvqk, w.t())


def _cast(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """A parameter in the activations' dtype.  The fused nodes take the module's parameters AS THEY ARE (fp32 masters next
    to bf16 activations): one cast per parameter and forward inside the node, the casted copy saved for backward, and the
    gradient -- which the kernels and the split weight-gradient GEMM produce in fp32 anyway -- returned in the
    parameter's dtype without a detour through bf16 (a ``.to()`` outside the node costs a second cast kernel in forward's
    recompute, one more in backward, and rounds the gradient to bf16 on the way)."""
    return t if t.dtype == dtype else t.detach().to(dtype)


class _NormMulFunction(torch.autograd.Function):
    """y = dropout(u * Norm(attn)) (optionally [u, attn, y]) as a node of its own, without the GEMM: the research layer's
    output stage (its projection is an nn.Linear) and the tests' handle on the row kernels.  Dropout is fused exactly as in
    _ComputeOutputFunction: the mask is regenerated from the seed in backward."""

    @staticmethod
    def forward(ctx, attn, u, weight, bias, eps, num_heads, linear_dim, group_norm, concat_ux, dropout_ratio=0.0, seed=0):
        y, mean, rstd = _launch.norm_mul_fwd(attn, u, weight, bias, eps, num_heads, linear_dim, group_norm, concat_ux,
                                             dropout_ratio, seed)
        ctx.save_for_backward(attn, u, weight, bias, mean, rstd)
        ctx.meta = (num_heads, linear_dim, group_norm, concat_ux, dropout_ratio, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        attn, u, weight, bias, mean, rstd = ctx.saved_tensors
        H, Ld, gn, cat, p_drop, seed = ctx.meta
        dattn, du, dw, db = _launch.norm_mul_bwd(dy, attn, u, weight, bias, mean, rstd, H, Ld, gn, cat, p_drop, seed)
        return dattn, du, dw.to(weight.dtype), db.to(bias.dtype), None, None, None, None, None, None, None


def _out_dgrad(dout, Wo):
    """d y = d out . W_out^T (reference: dx = torch.mm(dz, w.t()), ops/triton/triton_addmm.py:302-315).  The contraction runs over
    the embedding dim; at 512 the hand-written kernel takes it (W_out as stored IS its K-contiguous operand), else hipBLASLt.
    ``HSTU_OUT_DGRAD_KERNEL=0``: always hipBLASLt (A/B runs)."""
    if (os.environ.get("HSTU_OUT_DGRAD_KERNEL", "1") != "0" and Wo.is_contiguous() and Wo.dtype == dout.dtype
            and _launch.linear_k512_supported(dout, Wo.shape[0])):
        return _launch.linear_k512(dout, Wo)
    return torch.mm(dout, Wo.t())


def _residual_addmm(x, y, w):
    """out = x + y @ w: ONE hipBLASLt launch that reads x and writes out (hstu_addmm_residual, ABI v13) where the operands allow it,
    else torch.addmm -- the same library behind a copy of x into the result and an in-place GEMM"""
    if _launch.addmm_residual_supported(x, y, w):
        return _launch.addmm_residual(x, y, w)
    return torch.addmm(x, y, w)


class _ComputeOutputFunction(torch.autograd.Function):
    """out = x + [u, attn, u*Norm(attn)] @ W_o as ONE node (HSTUComputeOutputFunction,
    triton_hstu_linear.py:1137-1308): y is recomputed in backward unless asked otherwise."""

    @staticmethod
    def forward(ctx, attn, u, x, norm_weight, norm_bias, output_weight, eps, num_heads, linear_dim, concat_ux,
                group_norm, recompute_y, dropout_ratio=0.0, seed=0, grad_on=True):
        # dropout (training): inside the norm kernel, on all of [u, attn, u * Norm(attn)] as the reference's
        # _ln_mul_dropout_fwd does (triton_hstu_linear.py:101-120); the mask is never stored -- the backward kernel and the
        # recompute of y regenerate it from the seed
        ctx.param_dtypes = (norm_weight.dtype, norm_bias.dtype, output_weight.dtype)
        (norm_weight, norm_bias, output_weight), _ = _prepare_params((norm_weight, norm_bias, output_weight), x.dtype, None,
                                                                      grad_on and any(ctx.needs_input_grad))
        y, mean, rstd = _launch.norm_mul_fwd(attn, u, norm_weight, norm_bias, eps, num_heads, linear_dim, group_norm,
                                             concat_ux, dropout_ratio, seed)
        out = _residual_addmm(x, y, output_weight)
        saved = [attn, u, norm_weight, norm_bias, mean, rstd, output_weight]
        if not recompute_y:
            saved.append(y)
        ctx.save_for_backward(*saved)
        ctx.meta = (eps, num_heads, linear_dim, concat_ux, group_norm, recompute_y, dropout_ratio, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        eps, H, Ld, cat, gn, recompute_y, p_drop, seed = ctx.meta
        attn, u, nw, nb, mean, rstd, Wo = ctx.saved_tensors[:7]
        if recompute_y:
            y, _, _ = _launch.norm_mul_fwd(attn, u, nw, nb, eps, H, Ld, gn, cat, p_drop, seed)
        else:
            y = ctx.saved_tensors[7]
        dout = dout.contiguous()
        nw_dtype, nb_dtype, wo_dtype = ctx.param_dtypes
        dy = _out_dgrad(dout, Wo)
        dWo = weight_grad_mm(y, dout, out_dtype=wo_dtype)
        dattn, du, dnw, dnb = _launch.norm_mul_bwd(dy, attn, u, nw, nb, mean, rstd, H, Ld, gn, cat, p_drop, seed)
        return (dattn, du, dout, dnw.to(nw_dtype), dnb.to(nb_dtype), dWo, None, None, None, None, None, None, None,
                None, None)


def hstu_compute_output(
    attn: torch.Tensor,
    u: torch.Tensor,
    x: torch.Tensor,
    norm_weight: torch.Tensor,
    norm_bias: torch.Tensor,
    norm_eps: float,
    output_weight: torch.Tensor,
    num_heads: int,
    linear_dim: int,
    dropout_ratio: float,
    training: bool,
    concat_ux: bool,
    group_norm: bool,
    recompute_y_in_backward: bool,
    kernel: HammerKernel = HammerKernel.HIP,
) -> torch.Tensor:
    del kernel
    p_drop = float(dropout_ratio) if training else 0.0
    torch._assert(0.0 <= p_drop < 1.0, "dropout_ratio must be in [0, 1)")
    seed = draw_dropout_seed() if p_drop > 0.0 else 0
    return _ComputeOutputFunction.apply(attn, u, x, norm_weight, norm_bias, output_weight, norm_eps, num_heads,
                                        linear_dim, concat_ux, group_norm, recompute_y_in_backward, p_drop, seed,
                                        torch.is_grad_enabled())


class _PreprocessAndAttentionFunction(torch.autograd.Function):
    """LN -> UVQK GEMM -> split -> SiLU(u) -> jagged attention as one autograd node
    (_HSTUPreprocessAndAttentionFunction, triton_hstu_preprocess_and_attention.py:37-293).
    q, k, v are strided VIEWS of the fused uvqk buffer (never copied); in backward the
    attention kernel writes dq, dk, dv straight into the slices of one duvqk buffer."""

    @staticmethod
    def forward(ctx, x, norm_weight, norm_bias, uvqk_weight, uvqk_bias, seq_offsets, num_targets, norm_eps,
                num_heads, attn_dim, hidden_dim, max_seq_len, attn_alpha, max_attn_len, contextual_seq_len,
                recompute_uvqk, recompute_normed_x, user_order=None, grad_on=True):
        ctx.param_dtypes = (norm_weight.dtype, norm_bias.dtype, uvqk_weight.dtype, uvqk_bias.dtype)
        (norm_weight, norm_bias, uvqk_weight, uvqk_bias), ctx.kmajor = _prepare_params(
            (norm_weight, norm_bias, uvqk_weight, uvqk_bias), x.dtype, 2, grad_on and any(ctx.needs_input_grad))
        uvqk, normed_x, mean, rstd = _ln_uvqk(x, norm_weight, norm_bias, norm_eps, uvqk_weight, ctx.kmajor, uvqk_bias,
                                              want_normed=not recompute_normed_x)
        hv, ha = hidden_dim * num_heads, attn_dim * num_heads
        u_pre = uvqk[:, :hv]
        v = uvqk[:, hv : 2 * hv].view(-1, num_heads, hidden_dim)
        q = uvqk[:, 2 * hv : 2 * hv + ha].view(-1, num_heads, attn_dim)
        k = uvqk[:, 2 * hv + ha :].view(-1, num_heads, attn_dim)
        u = _launch.silu_fwd(u_pre)
        out = _launch.attn_fwd(q, k, v, seq_offsets, num_targets, max_seq_len, attn_alpha, 1.0 / max_seq_len,
                               max_attn_len, contextual_seq_len, 0, user_order=user_order)
        ctx.user_order = user_order
        saved = [x, norm_weight, norm_bias, mean, rstd, uvqk_weight, uvqk_bias, seq_offsets]
        ctx.has_targets = num_targets is not None
        if ctx.has_targets:
            saved.append(num_targets)
        ctx.keep_normed = not recompute_normed_x
        ctx.keep_uvqk = not recompute_uvqk
        if ctx.keep_normed:
            saved.append(normed_x)
        if ctx.keep_uvqk:
            saved.append(uvqk)
        ctx.save_for_backward(*saved)
        ctx.meta = (norm_eps, num_heads, attn_dim, hidden_dim, max_seq_len, attn_alpha, max_attn_len,
                    contextual_seq_len)
        return u, out.view(-1, hv)

    @staticmethod
    def backward(ctx, du, dout):
        saved = list(ctx.saved_tensors)
        x, nw, nb, mean, rstd, W, beta, seq_offsets = saved[:8]
        rest = saved[8:]
        num_targets = rest.pop(0) if ctx.has_targets else None
        normed_x = rest.pop(0) if ctx.keep_normed else None
        uvqk = rest.pop(0) if ctx.keep_uvqk else None
        eps, H, A, Hd, N, alpha, w, c = ctx.meta
        if uvqk is None:
            uvqk, normed_x = _recompute_uvqk(x, nw, nb, eps, W, ctx.kmajor, beta, normed_x)
        elif normed_x is None:
            normed_x, _, _ = _launch.layer_norm_fwd(x, nw, nb, eps)
        hv, ha = Hd * H, A * H
        v = uvqk[:, hv : 2 * hv].view(-1, H, Hd)
        q = uvqk[:, 2 * hv : 2 * hv + ha].view(-1, H, A)
        k = uvqk[:, 2 * hv + ha :].view(-1, H, A)
        duvqk = torch.empty_like(uvqk)
        dv = duvqk[:, hv : 2 * hv].view(-1, H, Hd)
        dq = duvqk[:, 2 * hv : 2 * hv + ha].view(-1, H, A)
        dk = duvqk[:, 2 * hv + ha :].view(-1, H, A)
        _launch.attn_bwd(dout.reshape(-1, H, Hd), q, k, v, seq_offsets, num_targets, N, alpha, 1.0 / N, w, c, 0,
                         dq=dq, dk=dk, dv=dv, user_order=ctx.user_order)
        _launch.silu_bwd(du, uvqk[:, :hv], din=duvqk[:, :hv])
        nw_dtype, nb_dtype, w_dtype, beta_dtype = ctx.param_dtypes
        dbeta = _bias_grad(duvqk, beta_dtype)
        d_normed = _uvqk_dgrad(duvqk, W, ctx.kmajor)
        dW = weight_grad_mm(normed_x, duvqk, out_dtype=w_dtype)
        dx, dnw, dnb = _launch.layer_norm_bwd(d_normed, x, nw, mean, rstd)
        return (dx, dnw.to(nw_dtype), dnb.to(nb_dtype), dW, dbeta.to(beta_dtype), None, None, None, None, None, None,
                None, None, None, None, None, None, None, None)


class _STULayerFunction(torch.autograd.Function):
    """One whole STU layer -- LN -> UVQK GEMM -> jagged attention -> dropout([u, attn, u * Norm(attn)]) @ W_o + x
    (stu.py:291-352 = hstu_preprocess_and_attention followed by hstu_compute_output) -- as ONE autograd node.  Compared
    with the two nodes above, which mirror the reference's pair of Triton functions, three row passes disappear:

    * SiLU(u): the output-stage kernels read the u slice of the uvqk buffer in place and apply SiLU on the fly, forward
      and (recomputed) in backward; SiLU' is applied where d u is stored, straight into the u slice of d uvqk.  No
      separate u tensor exists, so none is saved for backward either (one (sum L, H hidden) tensor per layer less).
    * the residual: the gradient that reaches x around the layer is added inside the layer-norm backward kernel instead
      of by autograd's accumulation of two gradients.

    Everything is rounded where the separate passes would have rounded it: with 16-bit activations outputs and gradients
    are bit-identical to the two-node path, in fp32 equal to an ulp or two
    (tests/test_compute_gpu.py::test_fused_layer_node_matches_two_nodes)."""

    @staticmethod
    def forward(ctx, x, in_nw, in_nb, uvqk_weight, uvqk_bias, out_nw, out_nb, output_weight, seq_offsets, num_targets,
                in_eps, out_eps, num_heads, attn_dim, hidden_dim, max_seq_len, attn_alpha, max_attn_len, contextual_seq_len,
                recompute_uvqk, recompute_normed_x, recompute_y, concat_ux, group_norm, dropout_ratio, seed, user_order,
                grad_on=True):
        ctx.param_dtypes = tuple(t.dtype for t in (in_nw, in_nb, uvqk_weight, uvqk_bias, out_nw, out_nb, output_weight))
        (in_nw, in_nb, uvqk_weight, uvqk_bias, out_nw, out_nb, output_weight), ctx.kmajor = _prepare_params(
            (in_nw, in_nb, uvqk_weight, uvqk_bias, out_nw, out_nb, output_weight), x.dtype, 2,
            grad_on and any(ctx.needs_input_grad))
        uvqk, normed_x, mean, rstd = _ln_uvqk(x, in_nw, in_nb, in_eps, uvqk_weight, ctx.kmajor, uvqk_bias,
                                              want_normed=not recompute_normed_x)
        hv, ha = hidden_dim * num_heads, attn_dim * num_heads
        v = uvqk[:, hv : 2 * hv].view(-1, num_heads, hidden_dim)
        q = uvqk[:, 2 * hv : 2 * hv + ha].view(-1, num_heads, attn_dim)
        k = uvqk[:, 2 * hv + ha :].view(-1, num_heads, attn_dim)
        attn = _launch.attn_fwd(q, k, v, seq_offsets, num_targets, max_seq_len, attn_alpha, 1.0 / max_seq_len,
                                max_attn_len, contextual_seq_len, 0, user_order=user_order).view(-1, hv)
        y, omean, orstd = _launch.norm_mul_fwd(attn, uvqk[:, :hv], out_nw, out_nb, out_eps, num_heads, hidden_dim, group_norm,
                                               concat_ux, dropout_ratio, seed, u_is_preactivation=True)
        out = _residual_addmm(x, y, output_weight)
        saved = [x, in_nw, in_nb, mean, rstd, uvqk_weight, uvqk_bias, seq_offsets, attn, out_nw, out_nb, omean, orstd,
                 output_weight]
        ctx.has_targets = num_targets is not None
        if ctx.has_targets:
            saved.append(num_targets)
        ctx.keep = (not recompute_normed_x, not recompute_uvqk, not recompute_y)
        for keep, t in zip(ctx.keep, (normed_x, uvqk, y)):
            if keep:
                saved.append(t)
        ctx.save_for_backward(*saved)
        ctx.user_order = user_order
        ctx.meta = (in_eps, out_eps, num_heads, attn_dim, hidden_dim, max_seq_len, attn_alpha, max_attn_len,
                    contextual_seq_len, concat_ux, group_norm, dropout_ratio, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        saved = list(ctx.saved_tensors)
        x, nw, nb, mean, rstd, W, beta, seq_offsets, attn, onw, onb, omean, orstd, Wo = saved[:14]
        rest = saved[14:]
        num_targets = rest.pop(0) if ctx.has_targets else None
        normed_x = rest.pop(0) if ctx.keep[0] else None
        uvqk = rest.pop(0) if ctx.keep[1] else None
        y = rest.pop(0) if ctx.keep[2] else None
        in_eps, out_eps, H, A, Hd, N, alpha, w, c, cat, gn, p_drop, seed = ctx.meta
        if uvqk is None:
            uvqk, normed_x = _recompute_uvqk(x, nw, nb, in_eps, W, ctx.kmajor, beta, normed_x)
        elif normed_x is None:
            normed_x, _, _ = _launch.layer_norm_fwd(x, nw, nb, in_eps)
        hv, ha = Hd * H, A * H
        u_pre = uvqk[:, :hv]
        if y is None:
            y, _, _ = _launch.norm_mul_fwd(attn, u_pre, onw, onb, out_eps, H, Hd, gn, cat, p_drop, seed,
                                           u_is_preactivation=True)
        dt = ctx.param_dtypes
        # ---- output stage (dout is also the gradient of the residual)
        dout = dout.contiguous()
        dy = _out_dgrad(dout, Wo)
        dWo = weight_grad_mm(y, dout, out_dtype=dt[6])
        del y
        duvqk = torch.empty_like(uvqk)
        dattn, _, donw, donb = _launch.norm_mul_bwd(dy, attn, u_pre, onw, onb, omean, orstd, H, Hd, gn, cat, p_drop, seed,
                                                    u_is_preactivation=True, du=duvqk[:, :hv])
        del dy
        # ---- attention
        v = uvqk[:, hv : 2 * hv].view(-1, H, Hd)
        q = uvqk[:, 2 * hv : 2 * hv + ha].view(-1, H, A)
        k = uvqk[:, 2 * hv + ha :].view(-1, H, A)
        dv = duvqk[:, hv : 2 * hv].view(-1, H, Hd)
        dq = duvqk[:, 2 * hv : 2 * hv + ha].view(-1, H, A)
        dk = duvqk[:, 2 * hv + ha :].view(-1, H, A)
        _launch.attn_bwd(dattn.view(-1, H, Hd), q, k, v, seq_offsets, num_targets, N, alpha, 1.0 / N, w, c, 0,
                         dq=dq, dk=dk, dv=dv, user_order=ctx.user_order)
        # ---- projections and the input norm (+ the residual's gradient, inside the kernel)
        dbeta = _bias_grad(duvqk, dt[3])
        d_normed = _uvqk_dgrad(duvqk, W, ctx.kmajor)
        dW = weight_grad_mm(normed_x, duvqk, out_dtype=dt[2])
        dx, dnw, dnb = _launch.layer_norm_bwd(d_normed, x, nw, mean, rstd, dresidual=dout)
        return (dx, dnw.to(dt[0]), dnb.to(dt[1]), dW, dbeta.to(dt[3]), donw.to(dt[4]), donb.to(dt[5]), dWo) + (None,) * 20


def hstu_fused_layer_applicable(x: torch.Tensor, attn_dim: int, hidden_dim: int) -> bool:
    """the single-node layer needs the head slices of the fused uvqk buffer 16-byte aligned (as the fused preprocess node)"""
    es = x.element_size()
    return x.is_cuda and (attn_dim * es) % 16 == 0 and (hidden_dim * es) % 16 == 0


def hstu_fused_layer(
    x: torch.Tensor,
    input_norm_weight: torch.Tensor,
    input_norm_bias: torch.Tensor,
    input_norm_eps: float,
    uvqk_weight: torch.Tensor,
    uvqk_bias: torch.Tensor,
    output_norm_weight: torch.Tensor,
    output_norm_bias: torch.Tensor,
    output_norm_eps: float,
    output_weight: torch.Tensor,
    num_heads: int,
    attn_dim: int,
    hidden_dim: int,
    max_seq_len: int,
    seq_offsets: torch.Tensor,
    attn_alpha: float,
    num_targets: Optional[torch.Tensor],
    max_attn_len: int,
    contextual_seq_len: int,
    dropout_ratio: float,
    training: bool,
    concat_ux: bool,
    group_norm: bool,
    recompute_uvqk_in_backward: bool,
    recompute_normed_x_in_backward: bool,
    recompute_y_in_backward: bool,
    sort_by_length: bool,
) -> torch.Tensor:
    """``hstu_compute_output(*hstu_preprocess_and_attention(x, ...)[:2], x, ...)`` -- the body of STULayer.forward
    (stu.py:291-352) -- as one autograd node (_STULayerFunction): same arguments, same results.  Not in the
    reference's API; STULayer uses it when no K/V has to be handed to the cache."""
    torch._assert(max_seq_len > 0, "max_seq_len must be larger than 0")
    torch._assert(x.dim() == 2, "x must be 2-D")
    torch._assert(x.shape[1] == uvqk_weight.shape[0], "x.shape[1] must equal uvqk_weight.shape[0]")
    torch._assert(
        uvqk_weight.shape[1] == 2 * num_heads * (hidden_dim + attn_dim),
        "uvqk_weight.shape[1] must equal 2 * num_heads * (hidden_dim + attn_dim)",
    )
    torch._assert(hstu_fused_layer_applicable(x, attn_dim, hidden_dim), "head dims must be 16-byte multiples")
    p_drop = float(dropout_ratio) if training else 0.0
    torch._assert(0.0 <= p_drop < 1.0, "dropout_ratio must be in [0, 1)")
    seed = draw_dropout_seed() if p_drop > 0.0 else 0
    order = _launch.length_order(_launch._idx(seq_offsets)) if sort_by_length and seq_offsets.numel() > 2 else None
    return _STULayerFunction.apply(
        x, input_norm_weight, input_norm_bias, uvqk_weight, uvqk_bias, output_norm_weight, output_norm_bias, output_weight,
        seq_offsets, num_targets, input_norm_eps, output_norm_eps, num_heads, attn_dim, hidden_dim, max_seq_len, attn_alpha,
        max_attn_len, contextual_seq_len, recompute_uvqk_in_backward, recompute_normed_x_in_backward,
        recompute_y_in_backward, concat_ux, group_norm, p_drop, seed, order, torch.is_grad_enabled())


def hstu_preprocess_and_attention(
    x: torch.Tensor,
    norm_weight: torch.Tensor,
    norm_bias: torch.Tensor,
    norm_eps: float,
    num_heads: int,
    attn_dim: int,
    hidden_dim: int,
    uvqk_weight: torch.Tensor,
    uvqk_bias: torch.Tensor,
    max_seq_len: int,
    seq_offsets: torch.Tensor,
    attn_alpha: float,
    causal: bool,
    num_targets: Optional[torch.Tensor],
    max_attn_len: int,
    contextual_seq_len: int,
    recompute_uvqk_in_backward: bool,
    recompute_normed_x_in_backward: bool,
    sort_by_length: bool,
    prefill: bool = False,
    kernel: HammerKernel = HammerKernel.HIP,
) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
    torch._assert(max_seq_len > 0, "max_seq_len must be larger than 0")
    torch._assert(x.dim() == 2, "x must be 2-D")
    torch._assert(x.shape[1] == uvqk_weight.shape[0], "x.shape[1] must equal uvqk_weight.shape[0]")
    torch._assert(
        uvqk_weight.shape[1] == 2 * num_heads * (hidden_dim + attn_dim),
        "uvqk_weight.shape[1] must equal 2 * num_heads * (hidden_dim + attn_dim)",
    )
    torch._assert(causal is True, "only causal attention is supported.")
    es = x.element_size()
    fusable = (attn_dim * es) % 16 == 0 and (hidden_dim * es) % 16 == 0
    if not prefill and fusable:
        u, attn_output = _PreprocessAndAttentionFunction.apply(
            x, norm_weight, norm_bias, uvqk_weight, uvqk_bias, seq_offsets, num_targets, norm_eps, num_heads,
            attn_dim, hidden_dim, max_seq_len, attn_alpha, max_attn_len, contextual_seq_len,
            recompute_uvqk_in_backward, recompute_normed_x_in_backward,
            _launch.length_order(_launch._idx(seq_offsets)) if sort_by_length and seq_offsets.numel() > 2 else None,
            torch.is_grad_enabled(),
        )
        return u, attn_output, None, None
    # prefill (k, v are returned for the KV cache) or head dims that need padding
    u, q, k, v = hstu_compute_uqvk(x, norm_weight, norm_bias, norm_eps, num_heads, attn_dim, hidden_dim,
                                   uvqk_weight, uvqk_bias)
    attn_output = hstu_mha(
        max_seq_len=max_seq_len, alpha=attn_alpha, q=q, k=k, v=v, seq_offsets=seq_offsets, causal=causal,
        dropout_pr=0.0, training=False, num_targets=num_targets, max_attn_len=max_attn_len,
        contextual_seq_len=contextual_seq_len, sort_by_length=sort_by_length,
    ).view(-1, hidden_dim * num_heads)
    return u, attn_output, k, v
