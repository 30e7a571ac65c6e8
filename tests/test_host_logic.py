"""Host-side logic that needs no GPU."""

import torch


def test_length_order_sorts_by_tile_count_and_keeps_batch_order_inside_a_tile_count():
    """sort_by_length's launch order (ops/_launch.py::length_order; reference: triton_hstu_attention.py:1968-1973 sorts by
    length): heavy users first at the granularity of the kernels' work -- 32-row tiles -- ties in batch order."""
    from generative_recommenders_amd.ops._launch import length_order

    lengths = torch.tensor([190, 200, 181, 193, 32, 33, 0, 199, 64, 1])
    off = torch.zeros(lengths.numel() + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lengths, 0)
    order = length_order(off).tolist()
    tiles = ((lengths + 31) // 32).tolist()
    assert sorted(order) == list(range(lengths.numel()))
    assert [tiles[i] for i in order] == sorted(tiles, reverse=True)
    for a, b in zip(order, order[1:]):
        assert tiles[a] > tiles[b] or a < b            # same tile count: batch order
    assert length_order(off) is length_order(off)      # cached per offsets tensor object


def test_addmm_residual_is_refused_for_what_the_one_launch_path_cannot_take():
    """ops/_launch.py::addmm_residual_supported (hstu_addmm_residual, ABI v13): CPU tensors, fp32, a 1-D bias, mismatched shapes and rows that
    are not 16-byte multiples all stay with torch.addmm -- decided on the host, before anything is launched."""
    from generative_recommenders_amd.ops import _launch

    x = torch.zeros(8, 16, dtype=torch.bfloat16)
    y = torch.zeros(8, 32, dtype=torch.bfloat16)
    w = torch.zeros(32, 16, dtype=torch.bfloat16)
    assert not _launch.addmm_residual_supported(x, y, w)                          # CPU
    assert not _launch.addmm_residual_supported(x.float(), y.float(), w.float())  # fp32
    assert not _launch.addmm_residual_supported(x[0], y, w)                       # 1-D input (a bias)
    assert not _launch.addmm_residual_supported(x[:, :8], y, w)                   # shape mismatch


def test_research_attention_orders_large_batches_only():
    from generative_recommenders_amd.research.modeling.sequential import hstu as R

    assert R._ORDER_MIN_USERS == 512
