"""Drop-in for generative_recommenders/ops/position.py:38-96: ``add_timestamp_positional_embeddings`` -- the step
right before the STU stack: ``alpha * seq_embeddings + position_table[pos index] + timestamp_table[time bucket]``
over jagged rows.  One HIP kernel forward (indices + gather + add), table gradients by a sorted segment sum
(csrc/position_ops.hip, csrc/embedding_grad.hip); semantics of ops/pytorch/pt_position.py:40-134."""

import ctypes as C
import os
from typing import Optional

import torch

from generative_recommenders_amd import _lib as L
from generative_recommenders_amd.common import HammerKernel
from generative_recommenders_amd.ops._launch import _idx

_FN = {"sqrt": 0, "log": 1}

# Largest time bucket.  The reference has two answers: its GPU path clamps to the last ROW of the timestamp table
# (ops/triton/triton_position.py:275,295: num_time_buckets = ts_emb.shape[0] - 1), its PyTorch path to
# ts_embeddings.size(1) - 1, the embedding DIM minus one (ops/pytorch/pt_position.py:101) -- with the shipped shapes
# (D = 512, 2048 buckets) the latter folds every bucket above 511 (time deltas beyond ~181 days) into one.  Checkpoints
# are trained on the GPU path, so "table" is the default; "pytorch_path" reproduces the PyTorch branch bit for bit
# (golden vectors: tests/golden/position.npz are minted from it).
TIME_BUCKET_CLAMP = os.environ.get("HSTU_TIME_BUCKET_CLAMP", "table")


def _max_time_bucket(ts_w: torch.Tensor, clamp: Optional[str] = None) -> int:
    clamp = clamp or TIME_BUCKET_CLAMP        # (module default: read at call time, so tests / callers may set it late)
    if clamp == "table":
        return ts_w.shape[0] - 1
    if clamp == "pytorch_path":
        return min(ts_w.shape[1] - 1, ts_w.shape[0] - 1)
    raise RuntimeError(f"time_bucket_clamp must be 'table' or 'pytorch_path', got {clamp!r}")


def _table_grad(g: torch.Tensor, idx: torch.Tensor, table_rows: int) -> torch.Tensor:
    """sum of the rows of g per table index -> (table_rows, D) fp32 (grouping and segment sum: csrc/embedding_grad.hip)"""
    n, dim = g.shape
    out = torch.empty((table_rows, dim), dtype=torch.float32, device=g.device)
    need = C.c_int64(0)
    L.check(L.lib().hstu_embedding_grad_workspace_bytes(n, table_rows, C.byref(need)))
    ws = torch.empty(max(need.value, 256), dtype=torch.uint8, device=g.device)
    with torch.cuda.device(g.device):
        L.check(L.lib().hstu_embedding_grad(g.data_ptr(), idx.data_ptr(), n, dim, table_rows, out.data_ptr(), ws.data_ptr(),
                                            ws.numel(), L.torch_dtype_code(g.dtype), L.current_stream_ptr(g.device)))
    return out


class _AddTsPosFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, x, seq_offsets, timestamps, num_targets, pos_w, ts_w, max_contextual_seq_len,
                interleave_targets, time_bucket_fn, time_bucket_clamp=None):
        for name, t in (("seq_embeddings", x), ("seq_offsets", seq_offsets), ("timestamps", timestamps),
                        ("position_embeddings_weight", pos_w), ("timestamp_embeddings_weight", ts_w)):
            L.require_gpu_tensor(t, name)
        x = x.contiguous()
        seq_offsets = _idx(seq_offsets)
        nt = None if num_targets is None else num_targets.to(seq_offsets.dtype).contiguous()
        ts = timestamps.to(torch.int64).contiguous()
        pw, tw = pos_w.detach().float().contiguous(), ts_w.detach().float().contiguous()
        rows, dim = x.shape
        out = torch.empty_like(x)
        pos_idx = torch.empty(rows, dtype=torch.int32, device=x.device)
        ts_idx = torch.empty(rows, dtype=torch.int32, device=x.device)
        max_bucket = _max_time_bucket(tw, time_bucket_clamp)
        if rows:
            with torch.cuda.device(x.device):
                L.check(L.lib().hstu_add_ts_pos_emb_fwd(
                    x.data_ptr(), out.data_ptr(), seq_offsets.data_ptr(), ts.data_ptr(), None if nt is None else nt.data_ptr(),
                    pw.data_ptr(), tw.data_ptr(), pos_idx.data_ptr(), ts_idx.data_ptr(), seq_offsets.numel() - 1, dim,
                    int(max_contextual_seq_len), pw.shape[0], max_bucket, int(bool(interleave_targets)),
                    _FN[time_bucket_fn], float(alpha), L.torch_dtype_code(x.dtype), L.index_dtype_code(seq_offsets),
                    L.current_stream_ptr(x.device)))
        ctx.save_for_backward(pos_idx, ts_idx)
        ctx.meta = (float(alpha), pos_w.shape[0], ts_w.shape[0], pos_w.dtype, ts_w.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        pos_idx, ts_idx = ctx.saved_tensors
        alpha, n_pos, n_ts, pos_dtype, ts_dtype = ctx.meta
        g = g.contiguous()
        dx = g * alpha
        dpos = _table_grad(g, pos_idx, n_pos).to(pos_dtype) if ctx.needs_input_grad[5] else None
        dts = _table_grad(g, ts_idx, n_ts).to(ts_dtype) if ctx.needs_input_grad[6] else None
        return None, dx, None, None, None, dpos, dts, None, None, None, None


def add_timestamp_positional_embeddings(
    alpha: float,
    max_seq_len: int,
    max_contextual_seq_len: int,
    position_embeddings_weight: torch.Tensor,
    timestamp_embeddings_weight: torch.Tensor,
    seq_offsets: torch.Tensor,
    seq_lengths: torch.Tensor,
    seq_embeddings: torch.Tensor,
    timestamps: torch.Tensor,
    num_targets: Optional[torch.Tensor],
    interleave_targets: bool,
    time_bucket_fn: str = "sqrt",
    kernel: HammerKernel = HammerKernel.HIP,
    time_bucket_clamp: Optional[str] = None,
) -> torch.Tensor:
    """Same signature as the reference (``max_seq_len`` / ``seq_lengths`` are implied by ``seq_offsets`` and unused:
    nothing is padded), plus ``time_bucket_clamp``: where the largest time bucket is clamped -- "table" (the last table
    row: the reference's GPU path, triton_position.py:275,295), "pytorch_path" (min(D - 1, last row): its PyTorch branch,
    pt_position.py:101), None = the module default TIME_BUCKET_CLAMP ("table" unless HSTU_TIME_BUCKET_CLAMP says
    otherwise)."""
    del max_seq_len, seq_lengths, kernel
    assert time_bucket_fn in ["sqrt", "log"]
    return _AddTsPosFunction.apply(alpha, seq_embeddings, seq_offsets, timestamps, num_targets, position_embeddings_weight,
                                   timestamp_embeddings_weight, max_contextual_seq_len, interleave_targets, time_bucket_fn,
                                   time_bucket_clamp)
