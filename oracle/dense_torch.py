"""Padded-dense torch restatement of the reference's PyTorch attention path, on CPU.
TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/hstu_oracle.py header for the rules).

This follows the reference's ALGORITHM step by step -- pad q, k, v to (B, H, N, d),
materialise the (B, H, N, N) logits, SiLU / N, multiply by the mask, second einsum, drop
the padding (ops/pytorch/pt_hstu_attention.py:87-171) -- so that timing it on the host
cores is a fair "what the reference's CPU path costs" number (bench.py cpu_baseline,
kind="port": /root/reference itself does not exist on the GPU box).  Backward is torch
autograd, as in the reference.  It is also cross-checked against the per-user numpy oracle
and the golden vectors in tests/test_oracle_golden.py.
"""

from typing import Optional

import torch
import torch.nn.functional as F


def _pad(values: torch.Tensor, offsets: torch.Tensor, n: int) -> torch.Tensor:
    """(sum L, H, d) -> (B, H, n, d), zero padded (role of fbgemm jagged_to_padded_dense)."""
    B = offsets.numel() - 1
    lengths = (offsets[1:] - offsets[:-1]).clamp(max=n)
    pos = torch.arange(n).view(1, n)
    mask = pos < lengths.view(B, 1)
    src = (offsets[:-1].view(B, 1) + pos)[mask]
    out = values.new_zeros((B * n,) + tuple(values.shape[1:]))
    out = out.index_put((torch.nonzero(mask.view(-1)).view(-1),), values.index_select(0, src))
    return out.view(B, n, values.shape[1], values.shape[2]).transpose(1, 2)


def _mask(n: int, lengths: torch.Tensor, num_targets: Optional[torch.Tensor], max_attn_len: int,
          contextual_seq_len: int, min_full_attn_seq_len: int) -> torch.Tensor:
    """(B or 1, n, n) bool; restates pt_hstu_attention.py:32-84 for causal attention."""
    ids = torch.arange(n).view(1, n)
    max_ids = lengths.view(-1, 1, 1)
    if contextual_seq_len > 0:
        ids = (ids - contextual_seq_len + 1).clamp(min=0)
        max_ids = max_ids - contextual_seq_len + 1
    if num_targets is not None:
        max_ids = max_ids - num_targets.view(-1, 1, 1)
        ids = torch.minimum(ids.view(1, 1, n), max_ids).view(-1, n)
    row = ids.view(-1, n, 1)
    col = ids.view(-1, 1, n)
    dist = row - col
    valid = torch.eye(n, dtype=torch.bool).view(1, n, n) | (dist > 0)
    if max_attn_len > 0:
        win = dist <= max_attn_len
        if min_full_attn_seq_len > 0:
            win = win | (row >= max_ids - min_full_attn_seq_len)
        valid = valid & win
    if contextual_seq_len > 0:
        valid = valid | ((row == 0) & (col < max_ids))
    return valid


def dense_hstu_mha(max_seq_len: int, alpha: float, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                   seq_offsets: torch.Tensor, num_targets: Optional[torch.Tensor] = None, max_attn_len: int = 0,
                   contextual_seq_len: int = 0, min_full_attn_seq_len: int = 0) -> torch.Tensor:
    L, H, _ = q.shape
    n = max_seq_len
    lengths = seq_offsets[1:] - seq_offsets[:-1]
    pq, pk, pv = _pad(q, seq_offsets, n), _pad(k, seq_offsets, n), _pad(v, seq_offsets, n)
    logits = torch.einsum("bhxa,bhya->bhxy", pq, pk) * alpha
    probs = F.silu(logits) / n
    probs = probs * _mask(n, lengths, num_targets, max_attn_len, contextual_seq_len, min_full_attn_seq_len).unsqueeze(1)
    dense = torch.einsum("bhxy,bhyv->bhxv", probs, pv).transpose(1, 2)  # (B, n, H, dv)
    B = lengths.numel()
    keep = (torch.arange(n).view(1, n) < lengths.clamp(max=n).view(B, 1)).view(-1)
    return dense.reshape(B * n, H, v.shape[2])[keep]


def dense_stu_layer(x: torch.Tensor, p: dict, num_heads: int, attn_dim: int, hidden_dim: int, max_seq_len: int,
                    seq_offsets: torch.Tensor, num_targets: Optional[torch.Tensor], group_norm: bool,
                    attn_alpha: Optional[float] = None, eps: float = 1e-6) -> torch.Tensor:
    """One STULayer forward on CPU with the reference's PyTorch-path algorithm (modules/stu.py:291-352):
    layer norm -> addmm(uvqk) -> split u|v|q|k, SiLU(u) (ops/hstu_compute.py:50-89) -> padded-dense attention (above)
    -> u * LN|GN(attn), concat [u, attn, y] (ops/pytorch/pt_hstu_linear.py:22-66) -> addmm(x, y, W_o) (:68-99).
    ``p`` holds the layer's parameters under the reference's names (``_input_norm_weight`` ...).  Backward is autograd."""
    H, A, Hd = num_heads, attn_dim, hidden_dim
    normed = F.layer_norm(x, (x.shape[1],), p["_input_norm_weight"], p["_input_norm_bias"], eps)
    uvqk = torch.addmm(p["_uvqk_beta"], normed, p["_uvqk_weight"])
    u, v, q, k = torch.split(uvqk, [Hd * H, Hd * H, A * H, A * H], dim=1)
    u = F.silu(u)
    alpha = attn_alpha if attn_alpha is not None else 1.0 / (A**0.5)
    attn = dense_hstu_mha(max_seq_len, alpha, q.reshape(-1, H, A), k.reshape(-1, H, A), v.reshape(-1, H, Hd), seq_offsets,
                          num_targets).reshape(-1, H * Hd)
    if group_norm:
        y = u * F.group_norm(attn.view(-1, H, Hd), num_groups=H, weight=p["_output_norm_weight"], bias=p["_output_norm_bias"],
                             eps=eps).view(-1, H * Hd)
    else:
        y = u * F.layer_norm(attn, (H * Hd,), p["_output_norm_weight"], p["_output_norm_bias"], eps)
    return torch.addmm(x, torch.cat([u, attn, y], dim=1), p["_output_weight"])


def dense_stu_stack(x: torch.Tensor, layers: list, **kw) -> torch.Tensor:
    """STUStack: the layers one after the other (modules/stu.py:421-466); ``layers`` = list of (params, group_norm)"""
    for prm, gn in layers:
        x = dense_stu_layer(x, prm, group_norm=gn, **kw)
    return x
