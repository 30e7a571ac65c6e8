"""Backend enum, module base class and synthetic length generators.

Mirrors the parts of generative_recommenders/common.py that callers of the op layer
touch: ``HammerKernel`` (:102-107), ``HammerModule`` (:110-170),
``generate_sparse_seq_len`` (:173-201), ``apply_sampling`` (:204-216).  This package has
ONE backend (the HIP library); the enum keeps the reference's members so existing call
sites type-check, and adds ``HIP``.  Whatever member is passed, the HIP kernels run --
there is no PyTorch/Triton path to dispatch to.
"""

import abc
from enum import Enum, unique
from typing import Any, Optional

import torch


@unique
class HammerKernel(Enum):
    TRITON = "TRITON"
    PYTORCH = "PYTORCH"
    CUDA = "CUDA"
    TRITON_CC = "TRITON_CC"
    HIP = "HIP"


class HammerModule(torch.nn.Module, abc.ABC):
    """Same surface as the reference's HammerModule; ``hammer_kernel()`` always answers HIP."""

    def __init__(self, is_inference: bool, training_dytpe: torch.dtype = torch.float32,
                 use_triton_cc: bool = True, hammer_kernel: Optional[HammerKernel] = None) -> None:
        super().__init__()
        self._is_inference = is_inference
        self._training_dtype = training_dytpe
        self._hammer_kernel = hammer_kernel
        self._use_triton_cc = use_triton_cc

    def hammer_kernel(self) -> HammerKernel:
        return HammerKernel.HIP

    def recursive_setattr(self, name: str, value: Any) -> None:
        for _, module in self.named_modules():
            if hasattr(module, name):
                setattr(module, name, value)

    def set_use_triton_cc(self, use_triton_cc: bool) -> None:
        self._use_triton_cc = use_triton_cc
        self.recursive_setattr("_use_triton_cc", use_triton_cc)

    def set_is_inference(self, is_inference: bool) -> None:
        self._is_inference = is_inference
        self.recursive_setattr("_is_inference", is_inference)

    def set_training_dtype(self, training_dtype: torch.dtype) -> None:
        self._training_dtype = training_dtype
        self.recursive_setattr("_training_dtype", training_dtype)

    def set_hammer_kernel(self, hammer_kernel: HammerKernel) -> None:
        self._hammer_kernel = hammer_kernel
        self.recursive_setattr("_hammer_kernel", hammer_kernel)

    @property
    def is_inference(self) -> bool:
        return self._is_inference

    @property
    def is_eval(self) -> bool:
        return (not self._is_inference) and (not self.training)

    @property
    def is_train(self) -> bool:
        return (not self._is_inference) and self.training


def generate_sparse_seq_len(size: int, max_seq_len: int, sparsity: float, device: torch.device) -> torch.Tensor:
    """Synthetic per-user lengths; same distribution family as the reference generator."""
    if sparsity == 0.0:
        return torch.zeros(size=(size,), device=device, dtype=torch.int)
    if sparsity == 1.0:
        return torch.full((size,), max_seq_len, device=device, dtype=torch.int)
    if sparsity >= 0.5:
        lo, hi = int((2 * sparsity - 1.0) * max_seq_len), max_seq_len
    else:
        lo, hi = 0, int(2 * sparsity * max_seq_len)
    return torch.randint(low=lo, high=hi, size=(size,), device=device, dtype=torch.int)


def apply_sampling(lengths: torch.Tensor, alpha: float, max_seq_len: int) -> torch.Tensor:
    """Stochastic-length subsampling: users longer than N^(alpha/2) are cut to it with
    probability 1 - N^alpha / L^2."""
    threshold = int(max_seq_len ** (alpha / 2))
    keep_prob = (max_seq_len**alpha) / torch.pow(lengths, 2)
    cut = torch.logical_and(lengths > threshold, torch.rand_like(keep_prob) < 1 - keep_prob)
    return torch.where(cut, threshold, lengths)
