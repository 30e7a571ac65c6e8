"""Drop-in for generative_recommenders/ops/mm.py:29-40.  The dense projections really are
GEMMs: they go to hipBLASLt through ``torch.addmm`` -- exactly what the reference itself
does on AMD (ops/hstu_compute.py:69-72, ops/triton/triton_hstu_linear.py:1191-1194)."""

import torch

from generative_recommenders_amd.common import HammerKernel


def addmm(input: torch.Tensor, mat1: torch.Tensor, mat2: torch.Tensor,
          kernel: HammerKernel = HammerKernel.HIP) -> torch.Tensor:
    del kernel
    return torch.addmm(input, mat1, mat2)
