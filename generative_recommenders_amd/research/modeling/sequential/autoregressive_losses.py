"""Drop-in for research/modeling/sequential/autoregressive_losses.py:30-256 of the reference: the negatives samplers
that feed ``SampledSoftmaxLoss`` (same class names, constructor arguments, ``process_batch`` / ``forward`` contracts),
plus the one thing the fused loss needs from them that the reference interface does not offer: ``sample_rows`` --
the draw WITHOUT gathering the (N', R, D) embedding tensor (the loss kernel gathers the rows itself).

Sampling (``torch.randint``) and the in-batch de-duplication (``torch.unique``) stay torch calls: index plumbing on a
few thousand ids, not the hot part.  ``BCELoss`` (:262-356) is not mirrored (one negative per row: nothing to fuse)."""

import abc
from typing import List, NamedTuple, Tuple

import torch


class SampledRows(NamedTuple):
    """What the fused loss consumes: ``table[rows]`` are the negatives' embeddings, ``ids`` their item ids."""
    rows: torch.Tensor          # (N', R) int64 indices into ``table``
    ids: torch.Tensor           # (N', R) int64 item ids of those rows
    table: torch.Tensor         # (X, D) embedding rows
    table_l2_norm: bool         # normalise gathered rows (False when ``table`` already holds normalised rows)


class NegativesSampler(torch.nn.Module):
    def __init__(self, l2_norm: bool, l2_norm_eps: float) -> None:
        super().__init__()
        self._l2_norm: bool = l2_norm
        self._l2_norm_eps: float = l2_norm_eps

    def normalize_embeddings(self, x: torch.Tensor) -> torch.Tensor:
        return self._maybe_l2_norm(x)

    def _maybe_l2_norm(self, x: torch.Tensor) -> torch.Tensor:
        if self._l2_norm:
            x = x / torch.clamp(torch.linalg.norm(x, ord=2, dim=-1, keepdim=True), min=self._l2_norm_eps)
        return x

    def _draw(self, shape, high: int, like: torch.Tensor) -> torch.Tensor:
        """The sampler's one random draw (autoregressive_losses.py:120-126, :190-196); overridable for tests."""
        return torch.randint(low=0, high=high, size=shape, dtype=like.dtype, device=like.device)

    @abc.abstractmethod
    def debug_str(self) -> str:
        pass

    @abc.abstractmethod
    def process_batch(self, ids: torch.Tensor, presences: torch.Tensor, embeddings: torch.Tensor) -> None:
        pass

    @abc.abstractmethod
    def sample_rows(self, positive_ids: torch.Tensor, num_to_sample: int) -> SampledRows:
        pass

    def forward(self, positive_ids: torch.Tensor, num_to_sample: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(sampled_ids, sampled_negative_embeddings) as in the reference: materialises (N', R, D)."""
        s = self.sample_rows(positive_ids, num_to_sample)
        emb = s.table[s.rows]
        return s.ids, (self.normalize_embeddings(emb) if s.table_l2_norm else emb)


class LocalNegativesSampler(NegativesSampler):
    """Uniform draws from the item catalogue (autoregressive_losses.py:73-131)."""

    def __init__(self, num_items: int, item_emb: torch.nn.Embedding, all_item_ids: List[int], l2_norm: bool,
                 l2_norm_eps: float) -> None:
        super().__init__(l2_norm=l2_norm, l2_norm_eps=l2_norm_eps)
        self._num_items: int = len(all_item_ids)
        self._item_emb: torch.nn.Embedding = item_emb
        self.register_buffer("_all_item_ids", torch.tensor(all_item_ids))

    def debug_str(self) -> str:
        return f"local{f'-l2-eps{self._l2_norm_eps}' if self._l2_norm else ''}"

    def process_batch(self, ids: torch.Tensor, presences: torch.Tensor, embeddings: torch.Tensor) -> None:
        pass

    def sample_rows(self, positive_ids: torch.Tensor, num_to_sample: int) -> SampledRows:
        shape = positive_ids.size() + (num_to_sample,)
        offsets = self._draw(shape, self._num_items, positive_ids)
        ids = self._all_item_ids[offsets.view(-1)].reshape(shape)
        return SampledRows(rows=ids, ids=ids, table=self._item_emb.weight, table_l2_norm=self._l2_norm)


class InBatchNegativesSampler(NegativesSampler):
    """Draws from the (optionally de-duplicated) items of the current batch (autoregressive_losses.py:134-204)."""

    def __init__(self, l2_norm: bool, l2_norm_eps: float, dedup_embeddings: bool) -> None:
        super().__init__(l2_norm=l2_norm, l2_norm_eps=l2_norm_eps)
        self._dedup_embeddings: bool = dedup_embeddings

    def debug_str(self) -> str:
        s = f"in-batch{f'-l2-eps{self._l2_norm_eps}' if self._l2_norm else ''}"
        return s + "-dedup" if self._dedup_embeddings else s

    def process_batch(self, ids: torch.Tensor, presences: torch.Tensor, embeddings: torch.Tensor) -> None:
        assert ids.size() == presences.size()
        assert ids.size() == embeddings.size()[:-1]
        if self._dedup_embeddings:
            valid_ids = ids[presences]
            unique_ids, inverse = torch.unique(input=valid_ids, sorted=False, return_inverse=True)
            offsets = torch.empty((unique_ids.numel(),), dtype=torch.int64, device=unique_ids.device)
            offsets[inverse] = torch.arange(valid_ids.numel(), dtype=torch.int64, device=unique_ids.device)
            self._cached_embeddings = self._maybe_l2_norm(embeddings[presences][offsets, :])
            self._cached_ids = unique_ids
        else:
            self._cached_embeddings = self._maybe_l2_norm(embeddings[presences])
            self._cached_ids = ids[presences]

    def get_all_ids_and_embeddings(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._cached_ids, self._cached_embeddings

    def sample_rows(self, positive_ids: torch.Tensor, num_to_sample: int) -> SampledRows:
        X = self._cached_ids.size(0)
        offsets = self._draw(positive_ids.size() + (num_to_sample,), X, positive_ids)
        # the cache holds normalised rows already (process_batch): the kernel uses them as they are
        return SampledRows(rows=offsets, ids=self._cached_ids[offsets], table=self._cached_embeddings, table_l2_norm=False)


class AutoregressiveLoss(torch.nn.Module):
    @abc.abstractmethod
    def jagged_forward(self, output_embeddings, supervision_ids, supervision_embeddings, supervision_weights,
                       negatives_sampler: NegativesSampler):
        pass

    @abc.abstractmethod
    def forward(self, lengths, output_embeddings, supervision_ids, supervision_embeddings, supervision_weights,
                negatives_sampler: NegativesSampler):
        pass
