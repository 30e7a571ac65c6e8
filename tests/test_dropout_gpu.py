"""Fused output-stage dropout (hstu_norm_mul_dropout_fwd / _bwd, ops/hstu_compute.py::hstu_compute_output with
training=True and dropout_ratio > 0; reference: triton_hstu_linear.py:49-128, 133-260, 1137-1308; pt_hstu_linear.py:23-99).

Parity under dropout has two legs, because the reference's Philox stream is not reproducible across implementations:
  (a) EXACT: the kernel's keep mask equals the oracle's restatement of the generator bit for bit, survivors are the
      no-dropout values times 65536 / (65536 - thr), the backward (and the recompute of y) use the SAME mask;
  (b) STATISTICAL: keep rate, independence of rows / columns / seeds -- what the reference's semantics promise."""

import numpy as np
import pytest
import torch

from oracle import hstu_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(rows, heads, hd, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    dim = heads * hd
    attn = torch.randn(rows, dim, generator=g).to(dtype).to(DEV)
    u = (torch.randn(rows, dim, generator=g) + 0.3).to(dtype).to(DEV)
    return attn, u


@pytest.mark.parametrize("group_norm", [True, False])
@pytest.mark.parametrize("concat", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,heads,hd", [(257, 4, 128), (33, 2, 24)])   # vectorised fast path / scalar general path
def test_mask_is_the_oracles_and_survivors_are_scaled(group_norm, concat, dtype, rows, heads, hd):
    from generative_recommenders_amd.ops import _launch

    attn, u = _inputs(rows, heads, hd, dtype)
    width = heads if group_norm else heads * hd
    w = (1 + 0.1 * torch.randn(width)).to(dtype).to(DEV)
    b = (0.5 + 0.1 * torch.randn(width)).to(dtype).to(DEV)       # bias away from 0: y == 0 only where it was dropped
    p, seed = 0.3, 0x1234_5678_9ABC_DEF1
    y0, m0, r0 = _launch.norm_mul_fwd(attn, u, w, b, 1e-5, heads, hd, group_norm, concat)
    y1, m1, r1 = _launch.norm_mul_fwd(attn, u, w, b, 1e-5, heads, hd, group_norm, concat, p, seed)
    assert torch.equal(m0, m1) and torch.equal(r0, r1)             # the statistics do not see the dropout
    keep, scale = O.dropout_keep_mask(seed, rows, y0.shape[1], p)
    keep_t = torch.from_numpy(keep).to(DEV)
    assert torch.equal(y1 != 0, keep_t & (y0 != 0)), "dropped elements are 0 and the mask is the oracle's"
    if dtype == torch.float32:
        want = torch.where(keep_t, y0 * np.float32(scale), torch.zeros_like(y0))
        assert torch.equal(y1, want), "kept elements = plain output x scale"
    else:   # the kernel scales in fp32 and rounds once; y0 has been rounded already: within one bf16 ulp of y0 x scale
        want = y0.float() * scale
        assert float(((y1.float() - want)[keep_t].abs() / want[keep_t].abs().clamp_min(1e-30)).max()) < 2 ** -7
    # same seed -> same tensor (the recompute of y in backward relies on it); another seed -> another mask
    y2, _, _ = _launch.norm_mul_fwd(attn, u, w, b, 1e-5, heads, hd, group_norm, concat, p, seed)
    assert torch.equal(y1, y2)
    y3, _, _ = _launch.norm_mul_fwd(attn, u, w, b, 1e-5, heads, hd, group_norm, concat, p, seed + 1)
    assert not torch.equal(y1, y3)


@pytest.mark.parametrize("group_norm", [True, False])
@pytest.mark.parametrize("concat", [True, False])
def test_backward_uses_the_forward_mask(group_norm, concat):
    """bwd(dy, seed) == bwd(dy * mask * scale, no dropout): the kernel regenerates the mask from the seed."""
    from generative_recommenders_amd.ops import _launch

    rows, heads, hd, dtype = 300, 4, 128, torch.float32
    attn, u = _inputs(rows, heads, hd, dtype, seed=3)
    width = heads if group_norm else heads * hd
    w = (1 + 0.1 * torch.randn(width)).to(DEV)
    b = (0.1 * torch.randn(width)).to(DEV)
    p, seed = 0.1, 987654321987
    y, mean, rstd = _launch.norm_mul_fwd(attn, u, w, b, 1e-5, heads, hd, group_norm, concat, p, seed)
    dy = torch.randn_like(y)
    got = _launch.norm_mul_bwd(dy, attn, u, w, b, mean, rstd, heads, hd, group_norm, concat, p, seed)
    keep, scale = O.dropout_keep_mask(seed, rows, y.shape[1], p)
    dy_masked = torch.where(torch.from_numpy(keep).to(DEV), dy * np.float32(scale), torch.zeros_like(dy))
    ref = _launch.norm_mul_bwd(dy_masked, attn, u, w, b, mean, rstd, heads, hd, group_norm, concat)
    for name, a, c in zip(("dattn", "du", "dweight", "dbias"), got, ref):
        assert torch.equal(a, c), name


def test_keep_rate_and_independence():
    from generative_recommenders_amd.ops import _launch

    rows, heads, hd = 4096, 4, 128
    attn, u = _inputs(rows, heads, hd, torch.bfloat16, seed=5)
    w = torch.ones(heads, dtype=torch.bfloat16, device=DEV)
    b = torch.full((heads,), 3.0, dtype=torch.bfloat16, device=DEV)
    u = u.abs() + 1                                                  # every plain output is non-zero
    attn = attn.abs() + 1
    for p in (0.1, 0.3, 0.5):
        y, _, _ = _launch.norm_mul_fwd(attn, u, w, b, 1e-5, heads, hd, True, True, p, 42)
        keep = (y != 0).float()
        n = keep.numel()
        thr = round(p * 65536)
        p_eff = thr / 65536
        sigma = (p_eff * (1 - p_eff) / n) ** 0.5
        assert abs(float(keep.mean()) - (1 - p_eff)) < 5 * sigma, (p, float(keep.mean()))
        # per column and per row: binomial with the same rate
        col = keep.mean(0)
        assert float((col - (1 - p_eff)).abs().max()) < 6 * (p_eff * (1 - p_eff) / rows) ** 0.5
        row = keep.mean(1)
        assert float((row - (1 - p_eff)).abs().max()) < 6 * (p_eff * (1 - p_eff) / keep.shape[1]) ** 0.5
        # neighbours (the two halves of one hash, adjacent hashes) and adjacent rows are uncorrelated
        k = keep - keep.mean()
        var = float((k * k).mean())
        for a, c in ((k[:, :-1], k[:, 1:]), (k[:, :-2], k[:, 2:]), (k[:-1], k[1:])):
            corr = float((a * c).mean()) / var
            assert abs(corr) < 5 / a.numel() ** 0.5, corr
        # another seed: an independent mask
        y2, _, _ = _launch.norm_mul_fwd(attn, u, w, b, 1e-5, heads, hd, True, True, p, 43)
        k2 = (y2 != 0).float() - keep.mean()
        assert abs(float((k * k2).mean()) / var) < 5 / n ** 0.5
        # survivors carry exactly one scale
        ratio = (y.float() / _launch.norm_mul_fwd(attn, u, w, b, 1e-5, heads, hd, True, True)[0].float())[keep.bool()]
        assert float((ratio - 65536 / (65536 - thr)).abs().max()) < 0.02    # bf16 rounding of the scaled value


@pytest.mark.parametrize("recompute_y", [True, False])
@pytest.mark.parametrize("group_norm", [True, False])
def test_compute_output_training_dropout_matches_masked_reference(monkeypatch, recompute_y, group_norm):
    """hstu_compute_output(training=True, dropout_ratio=p): out and every gradient equal the no-dropout pipeline with the
    oracle's mask applied to [u, attn, y] by hand (torch autograd on the GPU, fp32)."""
    from generative_recommenders_amd.ops import _launch, hstu_compute as hc

    rows, heads, hd, D = 200, 4, 32, 64
    dim = heads * hd
    g = torch.Generator().manual_seed(11)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    attn, u, x = mk(rows, dim).requires_grad_(), mk(rows, dim).requires_grad_(), mk(rows, D).requires_grad_()
    width = heads if group_norm else dim
    nw, nb = (1 + 0.1 * mk(width)).requires_grad_(), (0.1 * mk(width)).requires_grad_()
    Wo = (0.05 * mk(3 * dim, D)).requires_grad_()
    seed, p = 5550123, 0.25
    monkeypatch.setattr(hc, "draw_dropout_seed", lambda: seed)
    out = hc.hstu_compute_output(attn, u, x, nw, nb, 1e-5, Wo, heads, hd, p, True, True, group_norm, recompute_y)
    gout = torch.randn_like(out)
    out.backward(gout)
    got = [out.detach().clone()] + [t.grad.clone() for t in (attn, u, x, nw, nb, Wo)]
    for t in (attn, u, x, nw, nb, Wo):
        t.grad = None
    # reference: plain y3 (our own no-dropout node is already pinned against the reference: test_compute_gpu.py), masked by hand
    y3 = hc._NormMulFunction.apply(attn, u, nw, nb, 1e-5, heads, hd, group_norm, True)
    keep, scale = O.dropout_keep_mask(seed, rows, 3 * dim, p)
    y3 = y3 * torch.from_numpy(keep).to(DEV).float() * np.float32(scale)
    ref_out = torch.addmm(x, y3, Wo)
    ref_out.backward(gout)
    ref = [ref_out.detach()] + [t.grad for t in (attn, u, x, nw, nb, Wo)]
    for name, a, c in zip(("out", "dattn", "du", "dx", "dnorm_w", "dnorm_b", "dWo"), got, ref):
        rel = float((a - c).norm() / c.norm().clamp_min(1e-30))
        assert rel < 2e-6, (name, rel)
    # eval mode: no dropout, whatever the ratio
    e1 = hc.hstu_compute_output(attn, u, x, nw, nb, 1e-5, Wo, heads, hd, p, False, True, group_norm, recompute_y)
    e2 = hc.hstu_compute_output(attn, u, x, nw, nb, 1e-5, Wo, heads, hd, 0.0, True, True, group_norm, recompute_y)
    assert torch.equal(e1, e2)


def test_seed_comes_from_torchs_cpu_generator():
    from generative_recommenders_amd.ops import hstu_compute as hc

    torch.manual_seed(7)
    a = [hc.draw_dropout_seed() for _ in range(3)]
    torch.manual_seed(7)
    assert a == [hc.draw_dropout_seed() for _ in range(3)] and len(set(a)) == 3


def test_bad_ratio_is_refused():
    from generative_recommenders_amd.ops import _launch

    attn, u = _inputs(8, 2, 16, torch.float32)
    w, b = torch.ones(2, device=DEV), torch.zeros(2, device=DEV)
    with pytest.raises(RuntimeError, match="dropout_ratio"):
        _launch.norm_mul_fwd(attn, u, w, b, 1e-5, 2, 16, True, True, 1.0, 1)
