"""Operator seam: the reference's ``hstu::`` torch.library operators on MI355X, so that code calling ``torch.ops.hstu.*``
(cuda_hstu_attention.py:50, cuda_hstu_preprocess_and_attention.py:114,280, ops/benchmarks/hstu_attention_bench.py:263,290)
runs unchanged.

The schemas and their CUDA (== HIP tensors under PyTorch-ROCm) and Meta kernels are registered by a COMPILED library,
``libhstu_torch_ops.so`` (csrc/torch_ops/hstu_torch_ops.cpp, built by ``_lib.build()``), exactly as the reference's
extension does it (ops/cpp/hstu_attention/flash_api.cpp:275-365, flash_meta.cpp, ops/cpp/cpp_ops.cpp:94-135): a caller
needs nothing but

    torch.ops.load_library(generative_recommenders_amd.ops.torch_library.LIB_PATH)

``register()`` does that (idempotently).  Operators: hstu_mha / hstu_mha_fwd / hstu_mha_bwd (jagged or dense (B, S, H, d)
inputs, ``attn_scale`` read on the device as the reference's kernels do -- element 0 replaces 1/N), complete_cumsum,
expand_1d_jagged_to_dense, concat_1d_jagged_jagged, sort_kv_pairs.  Refused: fp8 descale tensors (no fp8
instantiation), head dims above 128, CPU tensors (the reference's CPU entries are dummies that return empty tensors,
flash_cpu_dummy.cpp; here the dispatcher raises).
"""

import os

import torch

from generative_recommenders_amd import _lib as L

LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libhstu_torch_ops.so")
_REGISTERED = False


def register() -> None:
    """Load the compiled operator library (once).  Raises if it has not been built, or if another library (e.g. the
    reference's CUDA extension) already owns the ``hstu`` schemas."""
    global _REGISTERED
    if _REGISTERED:
        return
    L.lib()                      # libhstu_hip.so first: the operator library links against it
    if not os.path.exists(LIB_PATH):
        raise L.HstuLibraryError(f"{LIB_PATH} not found: build it with generative_recommenders_amd._lib.build() "
                                 "(or __graft_entry__.build())")
    torch.ops.load_library(LIB_PATH)
    _REGISTERED = True
