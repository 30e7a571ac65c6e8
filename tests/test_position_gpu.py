"""GPU parity of the timestamp / position additive encoder through the C ABI: golden vectors of
the reference's PyTorch path, and the oracle on seeded inputs at sizes the oracle finishes in seconds.
Table indices are integers (position) / an fp32 bucket (time): compared EXACTLY.  Values: fp32 I/O within
1e-6 relative (one fused rounding instead of the reference's two), bf16 within bf16 rounding."""

import numpy as np
import pytest
import torch

from conftest import load_cases
from oracle import hstu_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _op():
    from generative_recommenders_amd.ops import position

    return position


@pytest.mark.parametrize("idx", range(3))
def test_golden_position_encoder_fwd_bwd(idx, monkeypatch):
    # the golden vectors come from the reference's PyTorch branch, whose time-bucket clamp is the embedding dim - 1
    # (pt_position.py:101); the op's default is the GPU branch's (last table row): case 0 (49 rows, D = 32) tells them apart
    monkeypatch.setattr(_op(), "TIME_BUCKET_CLAMP", "pytorch_path")
    c = load_cases("position.npz")[idx]
    x = torch.from_numpy(c["x"]).to(DEV).requires_grad_()
    pos_w = torch.from_numpy(c["pos_w"]).to(DEV).requires_grad_()
    ts_w = torch.from_numpy(c["ts_w"]).to(DEV).requires_grad_()
    nt = None if "num_targets" not in c else torch.from_numpy(c["num_targets"]).to(DEV)
    out = _op().add_timestamp_positional_embeddings(
        alpha=float(c["alpha"]), max_seq_len=int(c["N"]), max_contextual_seq_len=int(c["ctx"]),
        position_embeddings_weight=pos_w, timestamp_embeddings_weight=ts_w, seq_offsets=torch.from_numpy(c["offsets"]).to(DEV),
        seq_lengths=torch.from_numpy(c["lengths"]).to(DEV), seq_embeddings=x, timestamps=torch.from_numpy(c["ts"]).to(DEV),
        num_targets=nt, interleave_targets=bool(c["interleave"]), time_bucket_fn=str(c["fn"]))
    np.testing.assert_allclose(out.detach().cpu().numpy(), c["out"], rtol=1e-6, atol=1e-6)
    out.backward(torch.from_numpy(c["g"]).to(DEV))
    np.testing.assert_allclose(x.grad.cpu().numpy(), c["dx"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(pos_w.grad.cpu().numpy(), c["dpos_w"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ts_w.grad.cpu().numpy(), c["dts_w"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype,D,fn,targets,ctx,n_ts,clamp", [
    (torch.bfloat16, 512, "sqrt", True, 0, 200, "table"),        # clamp at the last table row (199 < sqrt buckets up to 288)
    (torch.float32, 64, "log", True, 4, 600, "table"),
    (torch.float16, 128, "sqrt", False, 0, 600, "table"),        # more buckets than the embedding dim: the GPU branch's clamp (599)
    (torch.float16, 128, "sqrt", False, 0, 600, "pytorch_path"),  # ... and the PyTorch branch's (127)
    (torch.bfloat16, 1024, "log", True, 2, 600, "table")])
def test_position_encoder_vs_oracle(dtype, D, fn, targets, ctx, n_ts, clamp, monkeypatch):
    """ragged lengths incl. empty and 1-row users, int32 offsets, segments far longer than one 256-row chunk of the
    gradient kernel (many rows share a time bucket), tables smaller than the sequence (index clamps), both time-bucket
    clamps of the reference (triton_position.py:275,295 vs pt_position.py:101)."""
    monkeypatch.setattr(_op(), "TIME_BUCKET_CLAMP", clamp)
    rng = np.random.default_rng(D + ctx)
    B, N = 37, 300
    lengths = rng.integers(ctx + 2, N + 1, size=B)
    lengths[3] = 0
    lengths[5] = ctx + 2
    nt = np.minimum(rng.integers(1, 6, size=B), np.maximum((lengths - ctx) // 2, 0)) if targets else None
    off = np.zeros(B + 1, dtype=np.int32)
    off[1:] = np.cumsum(lengths)
    Lt = int(off[-1])
    ts = np.concatenate([np.sort(rng.integers(0, 5 * 10**6, size=int(l))) for l in lengths]).astype(np.int64)
    n_pos = 128
    x = torch.from_numpy(rng.standard_normal((Lt, D))).to(dtype)
    pos_w = torch.from_numpy(rng.standard_normal((n_pos, D)) * 0.1).float()
    ts_w = torch.from_numpy(rng.standard_normal((n_ts, D)) * 0.1).float()
    g = torch.from_numpy(rng.standard_normal((Lt, D))).to(dtype)
    alpha = D**0.5
    xd, pw, tw = x.to(DEV).requires_grad_(), pos_w.to(DEV).requires_grad_(), ts_w.to(DEV).requires_grad_()
    out = _op().add_timestamp_positional_embeddings(
        alpha=alpha, max_seq_len=N, max_contextual_seq_len=ctx, position_embeddings_weight=pw, timestamp_embeddings_weight=tw,
        seq_offsets=torch.from_numpy(off).to(DEV), seq_lengths=torch.from_numpy(lengths).to(DEV), seq_embeddings=xd,
        timestamps=torch.from_numpy(ts).to(DEV), num_targets=None if nt is None else torch.from_numpy(nt).to(DEV),
        interleave_targets=False, time_bucket_fn=fn)
    k_pos, k_ts = (t.cpu().numpy() for t in out.grad_fn.saved_tensors)   # the kernel's table indices
    out.backward(g.to(DEV))
    ref, pos_idx, ts_idx = O.add_timestamp_positional_embeddings_fwd(
        alpha, x.double().numpy(), off, ts, pos_w.double().numpy(), ts_w.double().numpy(), ctx, nt, False, fn, bucket_clamp=clamp)
    # exact equality with the oracle's integer / fp32-bucket arithmetic
    assert np.array_equal(k_pos, pos_idx) and np.array_equal(k_ts, ts_idx)
    raw = max(int(O.time_bucket_indices(ts[off[b]:off[b + 1]], 10**9, fn).max()) for b in range(B) if lengths[b] > 0)
    limit = n_ts - 1 if clamp == "table" else min(D - 1, n_ts - 1)
    assert ts_idx.max() == min(raw, limit)
    if fn == "sqrt":
        assert raw > 199 and (raw > limit) == (n_ts == 200 or clamp == "pytorch_path")   # the clamp is exercised where intended
    rdx, rpos, rts = O.add_timestamp_positional_embeddings_bwd(alpha, g.double().numpy(), pos_idx, ts_idx, n_pos, n_ts)
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=1e-2)
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), ref, **tol)
    np.testing.assert_allclose(xd.grad.float().cpu().numpy(), rdx, **(dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=0.3)))
    np.testing.assert_allclose(pw.grad.cpu().numpy(), rpos, rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(tw.grad.cpu().numpy(), rts, rtol=2e-4, atol=2e-3)


def test_positional_encoder_module_names_and_cpu_error():
    from generative_recommenders_amd.modules.positional_encoder import HSTUPositionalEncoder

    m = HSTUPositionalEncoder(num_position_buckets=64, num_time_buckets=32, embedding_dim=16, contextual_seq_len=0)
    assert sorted(m.state_dict()) == ["_position_embeddings_weight", "_timestamp_embeddings_weight"]
    assert m._timestamp_embeddings_weight.shape == (33, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(max_seq_len=4, seq_lengths=torch.tensor([2]), seq_offsets=torch.tensor([0, 2]), seq_timestamps=torch.tensor([1, 2]),
          seq_embeddings=torch.zeros(2, 16), num_targets=None)


@pytest.mark.parametrize("dtype,dim,rows,n", [(torch.bfloat16, 512, 8192, 100_003), (torch.float32, 64, 37, 5_000), (torch.float16, 1024, 2049, 20_001),
                                              (torch.float32, 1024, 3, 777), (torch.bfloat16, 8, 1, 300), (torch.bfloat16, 512, 100, 0)])
def test_table_gradient_grouping_and_segment_sum(dtype, dim, rows, n):
    """csrc/embedding_grad.hip on its own: table row counts that are no power of two, one-row tables, segments cut by
    several 64-row runs and segments of one row, n = 0, 16-byte .. 4-KiB rows; the values are small integers, so every
    fp32 sum is exact whatever the order and the comparison is bit for bit.  The stable grouping makes a second call
    return the same bits for real-valued input too."""
    pos = _op()
    g = torch.Generator(device=DEV).manual_seed(n + rows)
    idx = torch.randint(0, rows, (n,), generator=g, device=DEV, dtype=torch.int32)
    if n > 1000:
        idx[100:700] = rows - 1                     # one long segment across run boundaries
    dout = torch.randint(-8, 9, (n, dim), generator=g, device=DEV).to(dtype)
    got = pos._table_grad(dout, idx, rows)
    ref = torch.zeros(rows, dim, dtype=torch.float32, device=DEV).index_add_(0, idx.long(), dout.float())
    assert torch.equal(got, ref)
    if n:
        real = torch.randn(n, dim, generator=g, device=DEV).to(dtype)
        a, b = pos._table_grad(real, idx, rows), pos._table_grad(real, idx, rows)
        ref = torch.zeros(rows, dim, dtype=torch.float64, device=DEV).index_add_(0, idx.long(), real.double())
        torch.testing.assert_close(a.double(), ref, rtol=1e-5, atol=1e-4)
        cut = torch.zeros(rows, dtype=torch.bool, device=DEV)      # rows whose segment no run boundary cuts are order-exact
        sidx = torch.sort(idx.long(), stable=True).values
        cut[sidx[63::64]] = True
        cut[sidx[64::64]] = True
        assert torch.equal(a[~cut], b[~cut])
