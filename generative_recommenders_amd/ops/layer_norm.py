"""Drop-in for ``layer_norm`` of generative_recommenders/ops/layer_norm.py:46-76 on the HIP
row kernels (fp32 math, affine; backward returns dx, dweight, dbias)."""

from typing import List, Optional

import torch

from generative_recommenders_amd.common import HammerKernel
from generative_recommenders_amd.ops import _launch


class _LayerNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y, mean, rstd = _launch.layer_norm_fwd(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.wdtype, ctx.bdtype = weight.dtype, bias.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd = ctx.saved_tensors
        dx, dw, db = _launch.layer_norm_bwd(dy, x, weight, mean, rstd)
        return dx, dw.to(ctx.wdtype), db.to(ctx.bdtype), None


def layer_norm(
    x: torch.Tensor,
    weight: torch.Tensor,
    bias: torch.Tensor,
    eps: float = 1e-5,
    kernel: HammerKernel = HammerKernel.HIP,
) -> torch.Tensor:
    del kernel
    shape = x.shape
    y = _LayerNormFunction.apply(x.reshape(-1, shape[-1]), weight, bias, eps)
    return y.view(shape)
