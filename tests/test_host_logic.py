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
