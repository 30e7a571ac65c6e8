"""Drop-in for generative_recommenders/ops/mm.py:29-40.  The dense projections really are
GEMMs: they go to hipBLASLt through ``torch.addmm`` -- exactly what the reference itself
does on AMD (ops/hstu_compute.py:69-72, ops/triton/triton_hstu_linear.py:1191-1194)."""

from typing import Optional

import torch

from generative_recommenders_amd.common import HammerKernel
from generative_recommenders_amd.ops import _launch


def addmm(input: torch.Tensor, mat1: torch.Tensor, mat2: torch.Tensor,
          kernel: HammerKernel = HammerKernel.HIP) -> torch.Tensor:
    del kernel
    # (a 2-D ``input`` of the result's shape -- the STU layer's residual -- without ATen's copy of it into the result; gradients
    # need the autograd-aware torch op)
    if not (torch.is_grad_enabled() and (input.requires_grad or mat1.requires_grad or mat2.requires_grad)) and \
            _launch.addmm_residual_supported(input, mat1, mat2):
        return _launch.addmm_residual(input, mat1, mat2)
    return torch.addmm(input, mat1, mat2)


_SPLIT_SLABS = 16          # measured (tools/sweep_wgrad.py, profiles/r02_sweep_wgrad.json): 16 slabs are the optimum for both
                           # projections from 194 K to 1.55 M rows (uvqk 1.04, output 0.97-1.06 PFLOP/s; 8: 0.96 / 0.72; 1: 0.48 / 0.37)
_MIN_SLAB_ROWS = 2048
_bmm_f32_ok = None


def _bmm_f32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """batched a @ b with an fp32 result (hipBLASLt accumulates in fp32 anyway; this keeps the partial sums of the
    split from being rounded to the activation dtype)."""
    global _bmm_f32_ok
    if _bmm_f32_ok is None:
        try:
            torch.bmm(a[:1, :8], b[:1, :, :8], out_dtype=torch.float32)
            _bmm_f32_ok = True
        except (TypeError, RuntimeError, NotImplementedError):
            _bmm_f32_ok = False
    if _bmm_f32_ok:
        return torch.bmm(a, b, out_dtype=torch.float32)
    return torch.bmm(a.float(), b.float())


def weight_grad_mm(x: torch.Tensor, dy: torch.Tensor, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``x^T dy`` for x (L, K), dy (L, N): the weight gradient of ``y = x W``.  The contraction runs over ALL jagged
    rows (L ~ 10^5..10^6) while the result is one small (K, N) matrix: as a single GEMM that is K N / (128 x 256)
    output tiles -- 32 workgroups on a 256-CU part for the DLRM-v3 projections.  The rows are therefore split into
    S = 16 slabs (fewer for short inputs: a slab keeps >= 2048 rows), multiplied as one batched GEMM (S x as many tiles) with fp32 partial results, and summed."""
    L, K = x.shape
    N = dy.shape[1]
    # slabs: 16 at the DLRM-v3 projection shapes (swept: profiles/r02_sweep_wgrad.json), fewer when the single GEMM already
    # has an output tile per CU (wide projections: the split only adds a reduction) or when the fp32 partials
    # (S x K x N x 4 bytes) would pass 64 MiB
    tiles = -(-K // 256) * -(-N // 256)
    S = min(_SPLIT_SLABS, L // _MIN_SLAB_ROWS, max(1, 256 // tiles), max(1, (64 << 20) // (K * N * 4)))
    out_dtype = out_dtype or x.dtype
    if not x.is_cuda:
        raise RuntimeError("weight_grad_mm: CPU tensors are not supported (one backend, no fallback: DESIGN.md 1)")
    if S <= 1:
        return torch.mm(x.t(), dy).to(out_dtype)
    slab = (L // S) // 64 * 64
    main = slab * S
    xs, dys = x[:main].view(S, slab, K).transpose(1, 2), dy[:main].view(S, slab, N)
    if main < L and _tail_slab_ok(x):
        # the rows behind the last whole slab: one more fp32 partial next to the slabs', summed with them (no ``.float()`` /
        # ``+=`` passes over the (K, N) result: three tiny kernels per weight gradient less)
        part = torch.empty((S + 1, K, N), dtype=torch.float32, device=x.device)
        torch.bmm(xs, dys, out_dtype=torch.float32, out=part[:S])
        torch.mm(x[main:].t(), dy[main:], out_dtype=torch.float32, out=part[S])
        return part.sum(dim=0).to(out_dtype)
    part = _bmm_f32(xs, dys)
    out = part.sum(dim=0)
    if main < L:
        out += torch.mm(x[main:].t(), dy[main:]).float()
    return out.to(out_dtype)


_tail_ok = None


def _tail_slab_ok(x: torch.Tensor) -> bool:
    """does this torch build take ``out_dtype`` together with ``out=`` for mm / bmm (probed once on tiny operands)"""
    global _tail_ok
    if _tail_ok is None:
        try:
            a = torch.ones(2, 16, 8, dtype=x.dtype, device=x.device)
            o = torch.empty(3, 8, 8, dtype=torch.float32, device=x.device)
            torch.bmm(a.transpose(1, 2), a, out_dtype=torch.float32, out=o[:2])
            torch.mm(a[0].t(), a[0], out_dtype=torch.float32, out=o[2])
            _tail_ok = bool((o == 16.0).all())
        except (TypeError, RuntimeError, NotImplementedError):
            _tail_ok = False
    return _tail_ok
