"""GPU parity of the short-sequence kernels (csrc/hstu_attn_solo.cuh: one wave per (user, head), max_seq_len <= 64, head
dims <= 32 -- the Amazon-Books shape of BASELINE.json's config 3) against the fp64 oracle on seeded long-tailed batches:
forward and backward, bf16 and fp16, head dims 8 / 16 / 24 / 32 (one instantiation, zero-filled staging), targets,
sliding window, contextual rows (forward: solo; backward: the general kernel takes those), edge lengths 0 / 1 / 32 / 33 /
64, strided q / k / v views of one fused buffer, heavy-first launch order.  Gates: tests/test_attention_gpu.py."""

import numpy as np
import pytest
import torch

from oracle import hstu_oracle as O
from test_attention_gpu import check_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _batch(rng, B, N, long_tail=True):
    lengths = rng.integers(0, min(30, N + 1), size=B) if long_tail else rng.integers(0, N + 1, size=B)
    lengths[rng.random(B) < 0.05] = N
    lengths[:6] = [N, 1, min(32, N), min(33, N), 0, max(N - 1, 0)]
    off = np.zeros(B + 1, dtype=np.int64)
    off[1:] = np.cumsum(lengths)
    return lengths, off


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d,N,H,targets,window,ctx", [(16, 61, 4, False, 0, 0), (32, 64, 2, True, 0, 0), (24, 50, 3, True, 7, 0),
                                                      (8, 33, 1, False, 0, 0), (16, 61, 2, True, 5, 3), (32, 40, 2, False, 0, 4)])
def test_short_sequences_fwd_bwd_vs_oracle(d, N, H, targets, window, ctx, dtype):
    from generative_recommenders_amd.ops import _launch
    from generative_recommenders_amd.ops.hstu_attention import hstu_mha

    rng = np.random.default_rng(1000 * d + N + ctx)
    B = 60
    lengths, off = _batch(rng, B, N)
    lengths = np.maximum(lengths, ctx) if ctx else lengths
    off = np.zeros(B + 1, dtype=np.int64)
    off[1:] = np.cumsum(lengths)
    nt = np.minimum(rng.integers(1, 5, size=B), np.maximum(lengths - ctx, 0)) if targets else None
    Lt = int(off[-1])
    q, k, v, do = (torch.from_numpy(rng.standard_normal((Lt, H, d)) * 0.5).to(dtype) for _ in range(4))
    assert _launch.attn_fwd_kernel_name(dtype, d, d, N, contextual_seq_len=ctx).startswith("hstu_attn_fwd_solo_kernel")
    assert _launch.attn_bwd_kernel_name(dtype, d, d, N, contextual_seq_len=ctx).startswith(
        "hstu_attn_bwd_solo_kernel" if ctx == 0 else "hstu_attn_bwd_kernel")
    qd, kd, vd = (t.to(DEV).requires_grad_() for t in (q, k, v))
    out = hstu_mha(N, d**-0.5, qd, kd, vd, torch.from_numpy(off).to(DEV), num_targets=None if nt is None else torch.from_numpy(nt).to(DEV),
                   max_attn_len=window, contextual_seq_len=ctx, sort_by_length=True)
    out.backward(do.to(DEV))
    args = (q.double().numpy(), k.double().numpy(), v.double().numpy(), off, nt, window, ctx)
    ref = O.hstu_mha_fwd(N, d**-0.5, *args)
    rq, rk, rv = O.hstu_mha_bwd(N, d**-0.5, do.double().numpy(), *args)
    check_close(out, ref, dtype, "out")
    check_close(qd.grad, rq, dtype, "dq")
    check_close(kd.grad, rk, dtype, "dk")
    check_close(vd.grad, rv, dtype, "dv")


def test_short_sequences_strided_views_and_both_kernel_families_agree():
    """q, k, v as views of one fused (L, H, 3d) buffer, dq / dk / dv written into views of another (the layer's layout);
    the result agrees with the general kernels' (HSTU_SOLO is read once per process, so the comparison is with the
    oracle-checked general path through a head dim of 33..64 -- here simply against the oracle again)."""
    from generative_recommenders_amd.ops import _launch

    rng = np.random.default_rng(5)
    B, N, H, d = 200, 61, 4, 16
    lengths, off = _batch(rng, B, N)
    Lt = int(off[-1])
    fused = torch.from_numpy(rng.standard_normal((Lt, H, 3 * d)) * 0.5).bfloat16().to(DEV)
    q, k, v = torch.split(fused, [d, d, d], dim=-1)
    do = torch.from_numpy(rng.standard_normal((Lt, H, d))).bfloat16().to(DEV)
    offd = torch.from_numpy(off).to(DEV)
    out = _launch.attn_fwd(q, k, v, offd, None, N, d**-0.5, 1.0 / N)
    dfused = torch.full_like(fused, float("nan"))
    dq, dk, dv = torch.split(dfused, [d, d, d], dim=-1)
    _launch.attn_bwd(do, q, k, v, offd, None, N, d**-0.5, 1.0 / N, dq=dq, dk=dk, dv=dv)
    assert torch.isfinite(dfused.float()).all()          # every gradient row was written
    args = (q.double().cpu().numpy(), k.double().cpu().numpy(), v.double().cpu().numpy(), off)
    check_close(out, O.hstu_mha_fwd(N, d**-0.5, *args), torch.bfloat16, "out")
    rq, rk, rv = O.hstu_mha_bwd(N, d**-0.5, do.double().cpu().numpy(), *args)
    check_close(dq, rq, torch.bfloat16, "dq")
    check_close(dk, rk, torch.bfloat16, "dk")
    check_close(dv, rv, torch.bfloat16, "dv")
