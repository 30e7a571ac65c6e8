"""Pure-torch stand-ins for the fbgemm_gpu ops the reference's PyTorch path
calls (fbgemm_gpu is not installed and there is no network).  Used ONLY by
``make_golden.py`` so the unmodified reference can run on CPU in the build
container.  CompositeImplicitAutograd, so autograd flows through them.
"""

from typing import List, Optional, Tuple

import torch

_lib = torch.library.Library("fbgemm", "DEF")
_lib.define(
    "jagged_to_padded_dense(Tensor values, Tensor[] offsets, SymInt[] max_lengths, float padding_value=0.0) -> Tensor"
)
_lib.define("dense_to_jagged(Tensor dense, Tensor[] x_offsets, SymInt? total_L=None) -> (Tensor, Tensor[])")
_lib.define("asynchronous_complete_cumsum(Tensor t_in) -> Tensor")
_lib.define("jagged_dense_elementwise_add_jagged_output(Tensor x_values, Tensor[] x_offsets, Tensor y) -> (Tensor, Tensor[])")


def _index_maps(offsets: torch.Tensor, max_len: int):
    lengths = (offsets[1:] - offsets[:-1]).clamp(max=max_len)
    B = lengths.numel()
    pos = torch.arange(max_len, device=offsets.device).view(1, -1)
    mask = pos < lengths.view(-1, 1)
    src = (offsets[:-1].view(-1, 1) + pos)[mask]
    return mask, src, B


def jagged_to_padded_dense(values, offsets: List[torch.Tensor], max_lengths: List[int], padding_value: float = 0.0):
    off, n = offsets[0], int(max_lengths[0])
    mask, src, B = _index_maps(off, n)
    trailing = values.shape[1:]
    out = values.new_full((B * n,) + tuple(trailing), padding_value)
    flat_dst = torch.nonzero(mask.view(-1)).view(-1)
    out = out.index_put((flat_dst,), values.index_select(0, src))
    return out.view((B, n) + tuple(trailing))


def dense_to_jagged(dense, x_offsets: List[torch.Tensor], total_L: Optional[int] = None) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    off = x_offsets[0]
    B, n = dense.shape[0], dense.shape[1]
    mask, _, _ = _index_maps(off, n)
    vals = dense.reshape((B * n,) + tuple(dense.shape[2:]))[mask.view(-1)]
    return vals, [off]


def asynchronous_complete_cumsum(t_in):
    out = t_in.new_zeros(t_in.numel() + 1)
    out[1:] = torch.cumsum(t_in, 0)
    return out


_lib.impl("jagged_to_padded_dense", jagged_to_padded_dense, "CompositeImplicitAutograd")
_lib.impl("dense_to_jagged", dense_to_jagged, "CompositeImplicitAutograd")
_lib.impl("asynchronous_complete_cumsum", asynchronous_complete_cumsum, "CompositeImplicitAutograd")


def jagged_dense_elementwise_add_jagged_output(x_values, x_offsets: List[torch.Tensor], y):
    """x_values (sum L, D) + y[b, position in user] (B, N, D), jagged output."""
    off = x_offsets[0]
    B, n = y.shape[0], y.shape[1]
    mask, _, _ = _index_maps(off, n)
    return x_values + y.reshape((B * n,) + tuple(y.shape[2:]))[mask.view(-1)], [off]


_lib.impl("jagged_dense_elementwise_add_jagged_output", jagged_dense_elementwise_add_jagged_output, "CompositeImplicitAutograd")
