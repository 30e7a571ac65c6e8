"""The post-processing step of ``HSTUTransducer`` (generative_recommenders/modules/hstu_transducer.py:191-251): split
the candidate (target) rows off every user's sequence and run the output postprocessor on them.  The preprocessors /
embedding tables in front of the transducer are control plane and are not mirrored; this function is the
``_postprocess`` method as a free function over the same arguments."""

from typing import Dict, Optional, Tuple

import torch

from generative_recommenders_amd.common import HammerKernel
from generative_recommenders_amd.modules.postprocessors import OutputPostprocessor
from generative_recommenders_amd.ops.jagged_tensors import asynchronous_complete_cumsum, split_2D_jagged


def hstu_postprocess(
    output_postprocessor: OutputPostprocessor,
    max_seq_len: int,
    total_uih_len: int,
    total_targets: int,
    seq_lengths: torch.Tensor,
    seq_timestamps: torch.Tensor,
    seq_embeddings: torch.Tensor,
    num_targets: torch.Tensor,
    seq_payloads: Dict[str, torch.Tensor],
    return_full_embeddings: bool = False,
    interleave_targets: bool = False,
    kernel: HammerKernel = HammerKernel.HIP,
) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    if return_full_embeddings:
        seq_embeddings = output_postprocessor(seq_embeddings=seq_embeddings, seq_timestamps=seq_timestamps,
                                              seq_payloads=seq_payloads)
    uih_offsets = asynchronous_complete_cumsum(seq_lengths - num_targets)
    candidates_offsets = asynchronous_complete_cumsum(num_targets)
    _, candidate_embeddings = split_2D_jagged(
        values=seq_embeddings, max_seq_len=max_seq_len, total_len_left=total_uih_len, total_len_right=total_targets,
        offsets_left=uih_offsets, offsets_right=candidates_offsets, kernel=kernel)
    if interleave_targets:
        candidate_embeddings = candidate_embeddings.view(-1, 2, candidate_embeddings.size(-1))[:, 0, :]
    if not return_full_embeddings:
        _, candidate_timestamps = split_2D_jagged(
            values=seq_timestamps.unsqueeze(-1), max_seq_len=max_seq_len, total_len_left=total_uih_len,
            total_len_right=total_targets, offsets_left=uih_offsets, offsets_right=candidates_offsets, kernel=kernel)
        candidate_timestamps = candidate_timestamps.squeeze(-1)
        if interleave_targets:
            candidate_timestamps = candidate_timestamps.view(-1, 2)[:, 0]
        candidate_embeddings = output_postprocessor(seq_embeddings=candidate_embeddings,
                                                    seq_timestamps=candidate_timestamps, seq_payloads=seq_payloads)
    return (seq_embeddings if return_full_embeddings else None), candidate_embeddings
