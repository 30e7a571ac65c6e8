"""A seeded slice of the randomised sweeps (tools/fuzz_attention.py, tools/fuzz_ops.py) inside the GPU suite.

The sweeps found round 2's only real bug (research forward, 129..160 rows, more than one head) and were not part of
`pytest -m gpu`; the full runs (thousands of cases, minutes of oracle time) stay tools, this is the minute of them the driver
executes every round: one case per test, fixed seeds, max_seq_len cycling through the tile-boundary lengths that have broken
kernels before, always at least two heads.  Every case is checked against the fp64 oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

# max_seq_len values on both sides of the kernels' tile / schedule boundaries (32-row tiles; 128-row forward blocks; the
# folded / 4-wave backward up to 224, several key blocks from 225 on), 129..160 being the range of the round-2 bug
BOUNDARY_N = [129, 146, 155, 160, 193, 224, 225]


@pytest.mark.parametrize("i", range(40))
def test_fuzz_hstu_mha(i):
    import fuzz_attention as F

    fails, _ = F.mha_sweep(1, seed=4000 + i, force_n=[BOUNDARY_N[i % len(BOUNDARY_N)]], exit_process=False)
    assert fails == 0


@pytest.mark.parametrize("i", range(6))
def test_fuzz_hstu_mha_free_shapes(i):
    """the sweep's own shape distribution (short one-wave shapes, several key blocks, mixed head dims), 5 cases per test"""
    import fuzz_attention as F

    fails, _ = F.mha_sweep(5, seed=4100 + i, exit_process=False)
    assert fails == 0


@pytest.mark.parametrize("i", range(30))
def test_fuzz_research_bias(i):
    import fuzz_attention as F

    fails, _ = F.bias_sweep(1, seed=4200 + i, force_n=[BOUNDARY_N[i % len(BOUNDARY_N)]], exit_process=False)
    assert fails == 0


@pytest.mark.parametrize("i", range(20))
def test_fuzz_delta_attention(i):
    import fuzz_ops as G

    assert G.run_slice(4300 + i, n_delta=1) == 0


@pytest.mark.parametrize("i", range(20))
def test_fuzz_row_ops(i):
    import fuzz_ops as G

    assert G.run_slice(4400 + i, n_row=1) == 0


@pytest.mark.parametrize("i", range(6))
def test_fuzz_jagged_movers_and_layer(i):
    import fuzz_ops as G

    assert G.run_slice(4500 + i, n_jagged=3, n_layer=1) == 0


@pytest.mark.parametrize("i", range(24))
def test_fuzz_ln_linear(i):
    """the fused LayerNorm + projection kernel (round 4) on random shapes: row counts on both sides of the 256-row block and of
    what one workgroup per CU covers (runs of units that start and end inside a block, more workgroups than units), column
    counts from one 32-column tile to the 4096 limit (odd tile counts included), both dtypes, with and without the bias, rows
    with a large common offset (the centred-variance pass), with and without the normalised rows as an output"""
    import numpy as np
    import torch

    from generative_recommenders_amd.ops import _launch
    from oracle import hstu_oracle as O

    rng = np.random.default_rng(7000 + i)
    rows = int(rng.choice([1, 31, 255, 256, 257, 511, 1023, 4097, 20000, 65537])) if i < 10 else int(rng.integers(1, 30000))
    n = 32 * int(rng.choice([1, 2, 3, 5, 16, 21, 64, 100, 128])) if i % 3 else 32 * int(rng.integers(1, 129))
    dtype = torch.bfloat16 if rng.random() < 0.6 else torch.float16
    shift = float(rng.choice([0.0, 0.0, 3.0, 30.0]))
    with_bias, want_normed = bool(rng.random() < 0.8), bool(rng.random() < 0.5)
    g = torch.Generator().manual_seed(7000 + i)
    x = (torch.randn(rows, 512, generator=g) * (0.3 + 2 * torch.rand(rows, 1, generator=g)) + shift * torch.randn(rows, 1, generator=g)).to(dtype)
    lw, lb = (1 + 0.2 * torch.randn(512, generator=g)).to(dtype), (0.2 * torch.randn(512, generator=g)).to(dtype)
    w = (torch.randn(512, n, generator=g) / 512**0.5).to(dtype)
    b = (0.2 * torch.randn(n, generator=g)).to(dtype) if with_bias else None
    y, normed, mean, rstd = _launch.ln_linear_fwd(x.cuda(), lw.cuda(), lb.cuda(), 1e-5, w.t().contiguous().cuda(), None if b is None else b.cuda(),
                                                  want_normed=want_normed)
    f = lambda t: t.double().numpy()
    nx = O.layer_norm_fwd(f(x), f(lw), f(lb), 1e-5)
    ref = nx @ f(w) + (f(b) if with_bias else 0.0)
    gate = 2.8e-3 if dtype == torch.bfloat16 else 3.2e-4
    rel = lambda a, r: float(np.linalg.norm(a.double().cpu().numpy() - r) / max(np.linalg.norm(r), 1e-30))
    case = dict(rows=rows, n=n, dtype=str(dtype), shift=shift, bias=with_bias, normed=want_normed)
    assert torch.isfinite(y).all(), case
    assert rel(y, ref) <= gate, (case, rel(y, ref))
    # row by row as well: one wrong row among 20,000 does not move a Frobenius norm
    rerr = np.linalg.norm(y.double().cpu().numpy() - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)
    assert rerr.max() <= 4 * gate, (case, int(rerr.argmax()), float(rerr.max()))
    if want_normed:
        assert rel(normed, nx) <= gate, case
    xs = f(x)
    np.testing.assert_allclose(mean.cpu().numpy(), xs.mean(axis=1), rtol=5e-6, atol=5e-6 * (1 + shift))
    np.testing.assert_allclose(rstd.cpu().numpy(), 1.0 / np.sqrt(xs.var(axis=1) + 1e-5), rtol=2e-5)
