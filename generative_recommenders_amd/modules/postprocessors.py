"""Drop-in for generative_recommenders/modules/postprocessors.py:30-103: the output postprocessors applied to the
(candidate) embeddings after the STU stack -- ``L2NormPostprocessor`` and ``LayerNormPostprocessor`` on the HIP row
kernels (same class / parameter names).  ``TimestampLayerNormPostprocessor`` (:106-197, a time-feature MLP in front
of the layer norm) is not on the path of the shipped configs and is not mirrored."""

from abc import abstractmethod
from typing import Dict

import torch

from generative_recommenders_amd.common import HammerModule
from generative_recommenders_amd.ops import _launch
from generative_recommenders_amd.ops.layer_norm import layer_norm


class _L2NormFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        ctx.save_for_backward(x)
        ctx.eps = eps
        return _launch.l2_norm_fwd(x, eps)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _launch.l2_norm_bwd(dy, x, ctx.eps), None


def l2_norm(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """x / max(||x||_2, eps) over the last dim"""
    return _L2NormFunction.apply(x, eps)


class OutputPostprocessor(HammerModule):
    """An abstract class for post-processing user embeddings after HSTU layers."""

    @abstractmethod
    def forward(self, seq_embeddings: torch.Tensor, seq_timestamps: torch.Tensor,
                seq_payloads: Dict[str, torch.Tensor]) -> torch.Tensor:
        pass


class L2NormPostprocessor(OutputPostprocessor):
    """Postprocesses user embeddings with l2 norm."""

    def __init__(self, is_inference: bool = False) -> None:
        super().__init__(is_inference=is_inference)

    def forward(self, seq_embeddings: torch.Tensor, seq_timestamps: torch.Tensor,
                seq_payloads: Dict[str, torch.Tensor]) -> torch.Tensor:
        return l2_norm(seq_embeddings, 1e-6)


class LayerNormPostprocessor(OutputPostprocessor):
    """Postprocesses user embeddings with layer norm."""

    def __init__(self, embedding_dim: int, eps: float = 1e-5, is_inference: bool = False) -> None:
        super().__init__(is_inference=is_inference)
        self._layer_norm: torch.nn.LayerNorm = torch.nn.LayerNorm(normalized_shape=[embedding_dim], eps=eps)

    def forward(self, seq_embeddings: torch.Tensor, seq_timestamps: torch.Tensor,
                seq_payloads: Dict[str, torch.Tensor]) -> torch.Tensor:
        ln = self._layer_norm
        return layer_norm(seq_embeddings.to(ln.weight.dtype), ln.weight, ln.bias, ln.eps)
