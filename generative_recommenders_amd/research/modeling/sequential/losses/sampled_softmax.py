"""Drop-in for research/modeling/sequential/losses/sampled_softmax.py:30-193 of the reference: ``SampledSoftmaxLoss``
with the dot-product similarity on one fused HIP gather + online-softmax kernel pair (``csrc/loss_ops.hip``).

Same constructor and the same two entry points: ``jagged_forward`` on (N', D) rows and ``forward`` on padded (B, N, D)
tensors + lengths (which only converts to jagged, with the jagged kernels of this package instead of fbgemm's).
``model`` is the object whose ``similarity_fn`` the reference calls; only the dot product is fused, so anything whose
``_ndp_module`` is not a dot-product similarity is refused (no silent fallback to a slow path)."""

from typing import Dict, Optional, Tuple

import torch

from generative_recommenders_amd.ops import _launch
from generative_recommenders_amd.research.modeling.sequential.autoregressive_losses import (
    AutoregressiveLoss,
    NegativesSampler,
)


class _SampledSoftmaxFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, pos_emb, table, pos_ids, neg_rows, neg_ids, temperature, pos_l2, table_l2, eps):
        row_loss, lse = _launch.sampled_softmax_fwd(q, pos_emb, pos_ids, neg_rows, neg_ids, table, temperature, pos_l2,
                                                    table_l2, eps)
        ctx.save_for_backward(q, pos_emb, table, pos_ids, neg_rows, neg_ids, lse)
        ctx.cfg = (temperature, pos_l2, table_l2, eps)
        return row_loss

    @staticmethod
    def backward(ctx, g_row):
        q, pos_emb, table, pos_ids, neg_rows, neg_ids, lse = ctx.saved_tensors
        temperature, pos_l2, table_l2, eps = ctx.cfg
        dq, dpos, dtable = _launch.sampled_softmax_bwd(g_row, lse, q, pos_emb, pos_ids, neg_rows, neg_ids, table,
                                                       temperature, pos_l2, table_l2, eps)
        return dq, dpos, dtable.to(table.dtype), None, None, None, None, None, None, None


def sampled_softmax_row_loss(output_embeddings: torch.Tensor, supervision_embeddings: torch.Tensor, table: torch.Tensor,
                             supervision_ids: torch.Tensor, neg_rows: torch.Tensor, neg_ids: torch.Tensor,
                             temperature: float, pos_l2_norm: bool, table_l2_norm: bool, eps: float) -> torch.Tensor:
    """Per-row ``-log_softmax([l_pos, l_neg_1 .. l_neg_R])[0]`` in fp32, differentiable w.r.t. the three embedding
    arguments (see include/hstu_hip.h: hstu_sampled_softmax_fwd/bwd)."""
    return _SampledSoftmaxFunction.apply(output_embeddings, supervision_embeddings, table, supervision_ids, neg_rows,
                                         neg_ids, float(temperature), bool(pos_l2_norm), bool(table_l2_norm), float(eps))


def _require_dot_product(model) -> None:
    ndp = getattr(model, "_ndp_module", None) if model is not None else None
    if model is None or ndp is None:
        return                                    # no similarity module given: the dot product is the definition here
    name = ndp.debug_str() if hasattr(ndp, "debug_str") else type(ndp).__name__
    if name != "dp" and "DotProduct" not in type(ndp).__name__:
        raise NotImplementedError(f"the fused sampled-softmax loss implements the dot-product similarity only, got {name}")


class SampledSoftmaxLoss(AutoregressiveLoss):
    def __init__(self, num_to_sample: int, softmax_temperature: float, model=None, activation_checkpoint: bool = False) -> None:
        super().__init__()
        _require_dot_product(model)
        self._num_to_sample: int = num_to_sample
        self._softmax_temperature: float = softmax_temperature
        self._model = model
        # accepted for signature compatibility: the fused op stores (rows, ids, lse) only -- there is no (N', R, D)
        # activation to checkpoint
        self._activation_checkpoint: bool = activation_checkpoint

    def jagged_forward(self, output_embeddings: torch.Tensor, supervision_ids: torch.Tensor,
                       supervision_embeddings: torch.Tensor, supervision_weights: torch.Tensor,
                       negatives_sampler: NegativesSampler, **kwargs) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        assert output_embeddings.size() == supervision_embeddings.size()
        assert supervision_ids.size() == supervision_embeddings.size()[:-1]
        assert supervision_ids.size() == supervision_weights.size()
        s = negatives_sampler.sample_rows(positive_ids=supervision_ids, num_to_sample=self._num_to_sample)
        table = s.table
        if table.dtype != output_embeddings.dtype:
            table = table.to(output_embeddings.dtype)       # autocast-style: parameters fp32, activations 16-bit
        row_loss = sampled_softmax_row_loss(
            output_embeddings, supervision_embeddings.to(output_embeddings.dtype), table, supervision_ids, s.rows, s.ids,
            self._softmax_temperature, negatives_sampler._l2_norm, s.table_l2_norm, negatives_sampler._l2_norm_eps)
        w = supervision_weights.to(row_loss.dtype)
        return (row_loss * w).sum() / w.sum(), {}

    def forward(self, lengths: torch.Tensor, output_embeddings: torch.Tensor, supervision_ids: torch.Tensor,
                supervision_embeddings: torch.Tensor, supervision_weights: torch.Tensor,
                negatives_sampler: NegativesSampler, **kwargs) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """lengths (B), output_embeddings / supervision_embeddings (B, N, D), supervision_ids / weights (B, N)."""
        from generative_recommenders_amd.ops.jagged_tensors import asynchronous_complete_cumsum as complete_cumsum, dense_to_jagged

        torch._assert(output_embeddings.size() == supervision_embeddings.size(), "Invalid supervision embeddings size.")
        torch._assert(supervision_ids.size() == supervision_embeddings.size()[:-1], "Invalid supervision ids size.")
        offsets = complete_cumsum(lengths)
        B, N = supervision_ids.shape
        # jagged row b*N + j for j < lengths[b]: one index vector serves every per-position tensor
        pos = torch.arange(N, device=lengths.device).unsqueeze(0)
        keep = (pos < lengths.unsqueeze(1)).reshape(-1)
        return self.jagged_forward(
            output_embeddings=dense_to_jagged(output_embeddings, offsets),
            supervision_ids=supervision_ids.reshape(-1)[keep],
            supervision_embeddings=dense_to_jagged(supervision_embeddings, offsets),
            supervision_weights=supervision_weights.reshape(-1)[keep],
            negatives_sampler=negatives_sampler, **kwargs)
