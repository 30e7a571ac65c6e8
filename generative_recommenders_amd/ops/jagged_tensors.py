"""Drop-in for generative_recommenders/ops/jagged_tensors.py (``concat_2D_jagged`` :55-90,
``split_2D_jagged`` :93-144, ``hstu_split_l2_embeddings`` :147-174,
``hstu_concat_l2_embeddings`` :177-207) plus the fbgemm-style helpers the hot path uses
(``jagged_to_padded_dense`` / ``dense_to_jagged`` / ``asynchronous_complete_cumsum``), all on
the HIP row-copy kernels (bit-exact).  Backward of a concat is the split of the gradient
and vice versa (cf. ops/triton/triton_jagged_tensors.py:145-359)."""

from typing import Optional, Tuple

import torch

from generative_recommenders_amd.common import HammerKernel
from generative_recommenders_amd.ops import _launch


def asynchronous_complete_cumsum(lengths: torch.Tensor) -> torch.Tensor:
    """[0, cumsum(lengths)], dtype preserved (fbgemm::asynchronous_complete_cumsum /
    hstu::complete_cumsum)."""
    return _launch.complete_cumsum(lengths)


complete_cumsum = asynchronous_complete_cumsum


class _ConcatFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values_left, values_right, offsets_left, offsets_right, max_len_left, max_len_right,
                max_seq_len, n_prefix):
        ctx.save_for_backward(*[t for t in (offsets_left, offsets_right) if t is not None])
        ctx.has = (offsets_left is not None, offsets_right is not None)
        ctx.meta = (max_len_left, max_len_right, max_seq_len, n_prefix, values_left.shape[0], values_right.shape[0])
        return _launch.concat_2d_jagged(values_left, values_right, offsets_left, offsets_right, max_len_left,
                                        max_len_right, max_seq_len, n_prefix)

    @staticmethod
    def backward(ctx, dout):
        saved = list(ctx.saved_tensors)
        ol = saved.pop(0) if ctx.has[0] else None
        orr = saved.pop(0) if ctx.has[1] else None
        mll, mlr, msl, npx, tl, tr = ctx.meta
        if ol is None and orr is None:  # both dense: synthesise the left offsets
            B = tl // mll
            ol = mll * torch.arange(B + 1, device=dout.device, dtype=torch.int64)
        dl, dr = _launch.split_2d_jagged(dout, tl, tr, ol, orr, mll, mlr, msl, npx)
        return dl, dr, None, None, None, None, None, None


class _SplitFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values, offsets_left, offsets_right, max_len_left, max_len_right, max_seq_len, n_prefix,
                total_left, total_right):
        ctx.save_for_backward(*[t for t in (offsets_left, offsets_right) if t is not None])
        ctx.has = (offsets_left is not None, offsets_right is not None)
        ctx.meta = (max_len_left, max_len_right, max_seq_len, n_prefix)
        return _launch.split_2d_jagged(values, total_left, total_right, offsets_left, offsets_right, max_len_left,
                                       max_len_right, max_seq_len, n_prefix)

    @staticmethod
    def backward(ctx, dleft, dright):
        saved = list(ctx.saved_tensors)
        ol = saved.pop(0) if ctx.has[0] else None
        orr = saved.pop(0) if ctx.has[1] else None
        mll, mlr, msl, npx = ctx.meta
        dv = _launch.concat_2d_jagged(dleft, dright, ol, orr, mll, mlr, msl, npx)
        return dv, None, None, None, None, None, None, None, None


def concat_2D_jagged(
    max_seq_len: int,
    values_left: torch.Tensor,
    values_right: torch.Tensor,
    max_len_left: Optional[int] = None,
    max_len_right: Optional[int] = None,
    offsets_left: Optional[torch.Tensor] = None,
    offsets_right: Optional[torch.Tensor] = None,
    kernel: HammerKernel = HammerKernel.HIP,
) -> torch.Tensor:
    torch._assert(values_left.dim() == 2, "values_left must be 2D")
    torch._assert(values_right.dim() == 2, "values_right must be 2D")
    torch._assert(
        values_right.shape[1] == values_left.shape[1],
        f"values_left shape[1] must be equal to values_right shape[1] {values_left.shape[1]} vs {values_right.shape[1]}",
    )
    if offsets_left is None:
        torch._assert(max_len_left is not None, "max_len_left must be provided when offsets_left is None")
    if offsets_right is None:
        torch._assert(max_len_right is not None, "max_len_right must be provided when offsets_right is None")
    del kernel
    return _ConcatFunction.apply(values_left, values_right, offsets_left, offsets_right, max_len_left,
                                 max_len_right, max_seq_len, 0)


def _side_total(offsets: Optional[torch.Tensor], max_len: Optional[int], other: torch.Tensor) -> int:
    if offsets is not None:
        return int(offsets[-1].item())  # host sync, as in the reference (_Split2DJaggedFunction :288)
    return int(max_len) * (other.shape[0] - 1)


def split_2D_jagged(
    max_seq_len: int,
    values: torch.Tensor,
    total_len_left: Optional[int] = None,
    total_len_right: Optional[int] = None,
    max_len_left: Optional[int] = None,
    max_len_right: Optional[int] = None,
    offsets_left: Optional[torch.Tensor] = None,
    offsets_right: Optional[torch.Tensor] = None,
    kernel: HammerKernel = HammerKernel.HIP,
) -> Tuple[torch.Tensor, torch.Tensor]:
    torch._assert(values.dim() == 2, "values must be 2D")
    torch._assert(
        offsets_left is not None or offsets_right is not None,
        "offsets_left and offsets_right cannot be None at the same time",
    )
    if offsets_left is None:
        torch._assert(max_len_left is not None, "max_len_left must be provided when offsets_left is None")
    if offsets_right is None:
        torch._assert(max_len_right is not None, "max_len_right must be provided when offsets_right is None")
    if offsets_left is not None and offsets_right is not None:
        torch._assert(offsets_left.shape[0] == offsets_right.shape[0],
                      "offsets_left shape[0] must be equal to offsets_right shape[0]")
    del kernel
    L = values.shape[0]
    if total_len_left is None and total_len_right is None:
        if offsets_left is not None:
            total_len_left = _side_total(offsets_left, None, offsets_left)
        else:
            total_len_left = _side_total(None, max_len_left, offsets_right)
        total_len_right = L - total_len_left
    elif total_len_left is None:
        total_len_left = L - total_len_right
    elif total_len_right is None:
        total_len_right = L - total_len_left
    return _SplitFunction.apply(values, offsets_left, offsets_right, max_len_left, max_len_right, max_seq_len, 0,
                                int(total_len_left), int(total_len_right))


def hstu_split_l2_embeddings(
    max_seq_len: int,
    x: torch.Tensor,
    prefix_offsets: torch.Tensor,
    l2_offsets: torch.Tensor,
    contextual_seq_len: int,
    kernel: HammerKernel = HammerKernel.HIP,
) -> Tuple[torch.Tensor, torch.Tensor]:
    del kernel
    total_prefix = int(prefix_offsets[-1].item())
    return _SplitFunction.apply(x, prefix_offsets, l2_offsets, None, None, max_seq_len, contextual_seq_len,
                                total_prefix, x.shape[0] - total_prefix)


def hstu_concat_l2_embeddings(
    max_prefix_len: int,
    prefix_x: torch.Tensor,
    prefix_offsets: torch.Tensor,
    max_l2_len: int,
    l2_x: torch.Tensor,
    l2_offsets: torch.Tensor,
    contextual_seq_len: int,
    kernel: HammerKernel = HammerKernel.HIP,
) -> torch.Tensor:
    del kernel
    return _ConcatFunction.apply(prefix_x, l2_x, prefix_offsets, l2_offsets, max_prefix_len, max_l2_len,
                                 max_prefix_len + max_l2_len, contextual_seq_len)


class _JaggedToPaddedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values, offsets, max_len):
        ctx.save_for_backward(offsets)
        ctx.total = values.shape[0]
        return _launch.jagged_to_padded_dense(values, offsets, max_len)

    @staticmethod
    def backward(ctx, ddense):
        (offsets,) = ctx.saved_tensors
        return _launch.dense_to_jagged(ddense, offsets, ctx.total), None, None


class _DenseToJaggedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dense, offsets, total_rows):
        ctx.save_for_backward(offsets)
        ctx.max_len = dense.shape[1]
        return _launch.dense_to_jagged(dense, offsets, total_rows)

    @staticmethod
    def backward(ctx, dvalues):
        (offsets,) = ctx.saved_tensors
        return _launch.jagged_to_padded_dense(dvalues, offsets, ctx.max_len), None, None


def jagged_to_padded_dense(values: torch.Tensor, offsets: torch.Tensor, max_length: int) -> torch.Tensor:
    """(sum L, ...) -> (B, max_length, ...), zero padded, rows >= max_length dropped."""
    return _JaggedToPaddedFunction.apply(values, offsets, int(max_length))


def dense_to_jagged(dense: torch.Tensor, offsets: torch.Tensor, total_L: Optional[int] = None) -> torch.Tensor:
    """(B, N, ...) -> (sum L, ...)."""
    if total_L is None:
        total_L = int(offsets[-1].item())
    return _DenseToJaggedFunction.apply(dense, offsets, int(total_L))


def expand_1d_jagged_to_dense(values: torch.Tensor, offsets: torch.Tensor, max_len: int) -> torch.Tensor:
    """hstu::expand_1d_jagged_to_dense (ops/cpp/cpp_ops.cpp:94-102)."""
    return _launch.expand_1d_jagged_to_dense(values, offsets, max_len)


def concat_1d_jagged_jagged(lengths_left, values_left, lengths_right, values_right) -> torch.Tensor:
    """hstu::concat_1d_jagged_jagged (ops/cpp/cpp_ops.cpp:94-102)."""
    return _launch.concat_1d_jagged_jagged(lengths_left, values_left, lengths_right, values_right)
