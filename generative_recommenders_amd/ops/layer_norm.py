"""Drop-in for generative_recommenders/ops/layer_norm.py on the HIP row kernels (fp32 math, affine; backward returns dx,
dweight, dbias): ``layer_norm`` (:46-76), ``swish_layer_norm`` (:79-112: x * sigmoid(LayerNorm(x))) and the two modules
that own the parameters, ``LayerNorm`` (:115-142) and ``SwishLayerNorm`` (:162-186).  (``RMSNorm`` :145-159 has no caller
in the reference and is not mirrored.)"""

from typing import List, Optional

import torch

from generative_recommenders_amd.common import HammerKernel, HammerModule
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


class _SwishLayerNormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y, mean, rstd = _launch.swish_layer_norm_fwd(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        dx, dw, db = _launch.swish_layer_norm_bwd(dy, x, weight, bias, mean, rstd)
        return dx, dw.to(weight.dtype), db.to(bias.dtype), None


def swish_layer_norm(
    x: torch.Tensor,
    weight: torch.Tensor,
    bias: torch.Tensor,
    eps: float = 1e-5,
    kernel: HammerKernel = HammerKernel.HIP,
) -> torch.Tensor:
    """``x * sigmoid(layer_norm(x))`` in one pass (reference: ops/layer_norm.py:79-112)."""
    del kernel
    shape = x.shape
    y = _SwishLayerNormFunction.apply(x.reshape(-1, shape[-1]), weight, bias, eps)
    return y.view(shape)


class LayerNorm(HammerModule):
    """ops/layer_norm.py:115-142: owns ``weight`` (ones) and ``bias`` (zeros) of shape (dim,)"""

    def __init__(self, dim: int, eps: float = 1e-5, is_inference: bool = False) -> None:
        super().__init__(is_inference=is_inference)
        self._normalized_shape: List[int] = [dim]
        self._eps = eps
        self.weight = torch.nn.Parameter(torch.ones(self._normalized_shape))
        self.bias = torch.nn.Parameter(torch.zeros(self._normalized_shape))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return layer_norm(x=x, weight=self.weight, bias=self.bias, eps=self._eps, kernel=self.hammer_kernel())


class SwishLayerNorm(HammerModule):
    """ops/layer_norm.py:162-186"""

    def __init__(self, dim: int, eps: float = 1e-5, is_inference: bool = False) -> None:
        super().__init__(is_inference=is_inference)
        self._normalized_shape: List[int] = [dim]
        self.weight = torch.nn.Parameter(torch.ones(self._normalized_shape))
        self.bias = torch.nn.Parameter(torch.zeros(self._normalized_shape))
        self._eps = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return swish_layer_norm(x=x, weight=self.weight, bias=self.bias, eps=self._eps, kernel=self.hammer_kernel())
