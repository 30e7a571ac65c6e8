"""GPU parity of the fused LayerNorm + UVQK projection kernel (csrc/hstu_ln_linear.cuh, C entry hstu_ln_linear_fwd)
against the numpy oracle (oracle/hstu_oracle.py::layer_norm_fwd + the GEMM of hstu_compute_uqvk, fp64) and against
the unfused product path (hstu_layer_norm_fwd + hipBLASLt) it replaces.

Tolerances (16-bit I/O, as for the unfused path in tests/test_compute_gpu.py): the oracle is exact arithmetic on the same
16-bit inputs; the kernel rounds LayerNorm(x) to the I/O dtype before the MFMA (as the reference's two kernels do through
memory) and y once more: relative Frobenius <= 2.8e-3 bf16 / 3.2e-4 fp16; mean / rstd fp32: 2e-6 relative."""

import numpy as np
import pytest
import torch

from conftest import record_parity
from oracle import hstu_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
GATE = {torch.bfloat16: 2.8e-3, torch.float16: 3.2e-4}


def _inputs(rows, n, dtype, seed, mean_shift=0.0, k=512):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(rows, k, generator=g) * (0.5 + torch.rand(rows, 1, generator=g)) + mean_shift * torch.randn(rows, 1, generator=g)).to(dtype)
    lw = (1 + 0.1 * torch.randn(k, generator=g)).to(dtype)
    lb = (0.1 * torch.randn(k, generator=g)).to(dtype)
    w = (torch.randn(k, n, generator=g) / k**0.5).to(dtype)        # the reference's (in, out) parameter
    b = (0.1 * torch.randn(n, generator=g)).to(dtype)
    return x, lw, lb, w, b


def _oracle(x, lw, lb, w, b, eps):
    f = lambda t: t.double().numpy()
    nx = O.layer_norm_fwd(f(x), f(lw), f(lb), eps)
    xs = f(x)
    mean = xs.mean(axis=1)
    rstd = 1.0 / np.sqrt(((xs - mean[:, None]) ** 2).mean(axis=1) + eps)
    return nx @ f(w) + f(b), nx, mean, rstd


def _fused(x, lw, lb, w, b, eps, want_normed=True):
    from generative_recommenders_amd.ops import _launch

    w_nk = w.t().contiguous().to(DEV)
    return _launch.ln_linear_fwd(x.to(DEV), lw.to(DEV), lb.to(DEV), eps, w_nk, None if b is None else b.to(DEV), want_normed=want_normed)


def _rel(got, ref):
    g = got.detach().double().cpu().numpy()
    return float(np.linalg.norm(g - ref) / max(np.linalg.norm(ref), 1e-30))


@pytest.mark.parametrize("rows,n,dtype", [(1, 32, torch.bfloat16), (31, 64, torch.float16), (256, 2048, torch.bfloat16),
                                          (257, 96, torch.bfloat16), (1000, 2048, torch.float16), (4099, 2048, torch.bfloat16),
                                          (70000, 512, torch.bfloat16), (66000, 2048, torch.bfloat16)])
def test_ln_linear_vs_oracle(rows, n, dtype):
    """row counts around the 256-row block and the 32-column tile, one workgroup and many, runs that start in the middle of a
    block (70000 rows x 16 tiles = 4384 units over 256 workgroups)"""
    eps = 1e-6
    x, lw, lb, w, b = _inputs(rows, n, dtype, seed=rows + n)
    y, normed, mean, rstd = _fused(x, lw, lb, w, b, eps)
    torch.cuda.synchronize()
    ry, rnx, rmean, rrstd = _oracle(x, lw, lb, w, b, eps)
    m = record_parity("ln_linear.y", y.double().cpu().numpy(), ry, str(dtype).replace("torch.", ""))
    assert m["rel_fro"] <= GATE[dtype], m
    assert _rel(normed, rnx) <= GATE[dtype]
    np.testing.assert_allclose(mean.cpu().numpy(), rmean, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(rstd.cpu().numpy(), rrstd, rtol=2e-6)
    assert torch.isfinite(y).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_ln_linear_rows_far_from_zero_mean(dtype):
    """|mean| up to ~100 sigma: the packed sum-of-squares form of the variance cancels, the wave takes the centred pass"""
    eps = 1e-6
    x, lw, lb, w, b = _inputs(777, 256, dtype, seed=5, mean_shift=20.0)
    y, normed, mean, rstd = _fused(x, lw, lb, w, b, eps)
    ry, rnx, rmean, rrstd = _oracle(x, lw, lb, w, b, eps)
    np.testing.assert_allclose(rstd.cpu().numpy(), rrstd, rtol=5e-6)
    assert _rel(y, ry) <= GATE[dtype]
    assert _rel(normed, rnx) <= GATE[dtype]


def test_ln_linear_matches_unfused_path_and_is_deterministic():
    """against hstu_layer_norm_fwd + torch's GEMM on the same inputs: the same roundings in the same places (normed_x to
    bf16, fp32 accumulation, y to bf16) -- differences are summation order only; and bit-identical from run to run"""
    from generative_recommenders_amd.ops import _launch

    eps = 1e-6
    x, lw, lb, w, b = _inputs(5000, 2048, torch.bfloat16, seed=11)
    y, normed, mean, rstd = _fused(x, lw, lb, w, b, eps)
    y2, _, _, _ = _fused(x, lw, lb, w, b, eps, want_normed=False)
    assert torch.equal(y, y2)
    nx, m0, r0 = _launch.layer_norm_fwd(x.to(DEV), lw.to(DEV), lb.to(DEV), eps)
    ref = torch.nn.functional.linear(nx, w.t().contiguous().to(DEV), b.to(DEV))
    torch.testing.assert_close(mean, m0, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(rstd, r0, rtol=2e-6, atol=0)
    # normed rows: equal up to the rare element whose fp32 value sits on a bf16 rounding boundary
    diff = (normed.float() - nx.float()).abs()
    assert (diff > 0).float().mean().item() < 2e-3, (diff > 0).float().mean().item()
    assert _rel(y, ref.double().cpu().numpy()) < 2.5e-3


def test_ln_linear_refuses_what_it_does_not_take():
    from generative_recommenders_amd.ops import _launch

    x = torch.randn(64, 256, device=DEV, dtype=torch.bfloat16)
    assert not _launch.ln_linear_supported(x, 512)                                         # k != 512
    assert not _launch.ln_linear_supported(torch.randn(64, 512, device=DEV), 512)        # fp32
    assert not _launch.ln_linear_supported(torch.randn(64, 512, device=DEV, dtype=torch.bfloat16), 100)
    with pytest.raises(RuntimeError, match="k == 512"):
        _launch.ln_linear_fwd(x, torch.ones(256, device=DEV), torch.zeros(256, device=DEV), 1e-6,
                              torch.randn(64, 256, device=DEV, dtype=torch.bfloat16), None)


@pytest.mark.parametrize("recompute", [True, False])
@pytest.mark.parametrize("fuse_layer", [True, False])
def test_stu_layer_with_fused_ln_uvqk_matches_two_kernel_path(recompute, fuse_layer):
    """an STU layer at the metric's width (embedding 512, 4 heads of 128, bf16 activations, fp32 master weights): output
    and every gradient with the fused LayerNorm + UVQK kernel (forward, and the recompute in backward) against the same
    layer on hstu_layer_norm_fwd + hipBLASLt (HSTU_LN_LINEAR=0) -- same roundings in the same places, only the summation
    order inside the row statistics and the GEMM differs: relative Frobenius 3e-3 per tensor (bf16 outputs), and against
    the fp32 layer within the bf16 gates the unfused path is held to"""
    from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig
    from generative_recommenders_amd.ops import hstu_compute

    D, H, Hd, A, N, B = 512, 4, 128, 128, 120, 9
    g = torch.Generator().manual_seed(23)
    lengths = torch.randint(1, N + 1, (B,), generator=g)
    off = torch.zeros(B + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lengths, 0)
    x0 = torch.randn(int(off[-1]), D, generator=g)
    gy = torch.randn(int(off[-1]), D, generator=g)
    torch.manual_seed(5)
    layer = STULayer(STULayerConfig(embedding_dim=D, num_heads=H, hidden_dim=Hd, attention_dim=A, output_dropout_ratio=0.0,
                                    causal=True, target_aware=True, max_attn_len=None, attn_alpha=None, use_group_norm=False,
                                    recompute_normed_x=recompute, recompute_uvqk=recompute, recompute_y=recompute,
                                    sort_by_length=False, contextual_seq_len=0)).to(DEV).train()
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g).to(DEV))
    layer.fuse_layer = fuse_layer
    res = {}
    old = hstu_compute._LN_LINEAR
    try:
        for fused in (True, False):
            hstu_compute._LN_LINEAR = fused
            layer.zero_grad(set_to_none=True)
            x = x0.to(DEV).to(torch.bfloat16).requires_grad_()
            y = layer(x=x, x_lengths=lengths.to(DEV), x_offsets=off.to(DEV), max_seq_len=N, num_targets=None)
            y.backward(gy.to(DEV).to(torch.bfloat16))
            res[fused] = [y.detach(), x.grad] + [p.grad for p in layer.parameters()]
    finally:
        hstu_compute._LN_LINEAR = old
    names = ["y", "dx"] + [n for n, _ in layer.named_parameters()]
    for n, a, b in zip(names, res[True], res[False]):
        assert a.dtype == b.dtype and a.shape == b.shape, n
        rel = float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))
        assert rel < 3e-3, f"{n}: {rel:.3e}"


def test_public_uvqk_op_runs_the_fused_kernel_forward_backward_and_row_results_do_not_depend_on_the_batch():
    """`hstu_compute_uqvk` (ops/hstu_compute.py:50-89) at embedding dim 512 with bf16 activations and fp32 master parameters:
    u, q, k, v and every gradient against fp64 autograd on the same bf16-representable inputs (relative Frobenius: q / k / v
    2.8e-3 = the bf16 gate of tests/test_compute_gpu.py, u 4.4e-3, gradients 8e-3: four bf16 roundings in the chain); and the rows of a
    sub-batch are bit-identical to the same rows computed inside the full batch (what makes K / V appended by a delta call
    equal to what the prefill wrote)"""
    from generative_recommenders_amd.ops.hstu_compute import hstu_compute_uqvk

    D, H, Hd, A, rows = 512, 4, 128, 128, 777
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(rows, D, generator=g).to(torch.bfloat16)
    nw0, nb0 = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    W0 = (torch.randn(D, 2 * H * (Hd + A), generator=g) / D**0.5)
    b0 = 0.1 * torch.randn(2 * H * (Hd + A), generator=g)
    # parameters as the kernel sees them (cast to bf16 inside the node): the fp64 reference starts from the same values
    rb = lambda t: t.to(torch.bfloat16).double()
    cot = [torch.randn(rows, H * Hd, generator=g), torch.randn(rows, H, A, generator=g), torch.randn(rows, H, A, generator=g),
           torch.randn(rows, H, Hd, generator=g)]

    def run(dev, dt, params):
        x = (x0.double() if dt == torch.float64 else x0).to(dev).requires_grad_()
        ps = [p.to(dev).requires_grad_() for p in params]
        if dt == torch.float64:
            nx = torch.nn.functional.layer_norm(x, (D,), ps[0], ps[1], 1e-6)
            uvqk = nx @ ps[2] + ps[3]
            u, v, q, k = torch.split(uvqk, [H * Hd, H * Hd, H * A, H * A], dim=1)
            outs = [torch.nn.functional.silu(u), q.reshape(-1, H, A), k.reshape(-1, H, A), v.reshape(-1, H, Hd)]
        else:
            outs = list(hstu_compute_uqvk(x, ps[0], ps[1], 1e-6, H, A, Hd, ps[2], ps[3]))
        loss = sum((o.double() * c.to(dev).double()).sum() for o, c in zip(outs, cot))
        loss.backward()
        return [o.detach() for o in outs], [x.grad] + [p.grad for p in ps]

    ref_o, ref_g = run("cpu", torch.float64, [rb(nw0), rb(nb0), rb(W0), rb(b0)])
    got_o, got_g = run(DEV, torch.bfloat16, [nw0, nb0, W0, b0])
    oerr = {name: _rel(a, b.numpy()) for name, a, b in zip(("u", "q", "k", "v"), got_o, ref_o)}
    assert all(a.dtype == torch.bfloat16 for a in got_o)
    # q, k, v: two bf16 roundings on the way (normed_x, the projection's output); u a third (SiLU's output): sqrt(3) x 1.66e-3
    assert max(oerr["q"], oerr["k"], oerr["v"]) <= 2.8e-3 and oerr["u"] <= 4.4e-3, oerr
    errs = {name: _rel(a, b.numpy()) for name, a, b in zip(("dx", "dnorm_weight", "dnorm_bias", "dW", "dbias"), got_g, ref_g)}
    assert all(e <= 8e-3 for e in errs.values()), errs
    assert got_g[3].dtype == torch.float32          # fp32 master weight: its gradient arrives in fp32
    with torch.no_grad():
        sub = torch.arange(40, 300, 7)
        full = hstu_compute_uqvk(x0.to(DEV), nw0.to(DEV), nb0.to(DEV), 1e-6, H, A, Hd, W0.to(DEV), b0.to(DEV))
        part = hstu_compute_uqvk(x0[sub].to(DEV), nw0.to(DEV), nb0.to(DEV), 1e-6, H, A, Hd, W0.to(DEV), b0.to(DEV))
        for f, p in zip(full, part):
            assert torch.equal(f[sub.to(DEV)], p)


# ---- hstu_linear_k512 (ABI v11): the same kernel without the LayerNorm -- d y = d out . W_out^T of the output stage's backward


@pytest.mark.parametrize("rows,n,dtype", [(1, 32, torch.bfloat16), (255, 1536, torch.float16), (257, 96, torch.bfloat16),
                                          (4099, 1536, torch.bfloat16), (70000, 1536, torch.bfloat16), (66000, 512, torch.float16)])
def test_linear_k512_vs_fp64_product(rows, n, dtype):
    """y = x W^T on the same 16-bit inputs in exact arithmetic (what the reference's dx = torch.mm(dz, w.t()) computes,
    ops/triton/triton_addmm.py:302-315): one rounding of y to the I/O dtype -- relative Frobenius <= 2.8e-3 bf16 / 3.2e-4 fp16 --
    and bit-identical from run to run; with and without a bias"""
    from generative_recommenders_amd.ops import _launch

    g = torch.Generator().manual_seed(rows + n)
    x = torch.randn(rows, 512, generator=g).to(dtype)
    w = (torch.randn(n, 512, generator=g) / 512**0.5).to(dtype)          # (n, k): `_output_weight` as stored
    b = (0.1 * torch.randn(n, generator=g)).to(dtype)
    xd, wd = x.to(DEV), w.to(DEV)
    assert _launch.linear_k512_supported(xd, n)
    y = _launch.linear_k512(xd, wd)
    yb = _launch.linear_k512(xd, wd, b.to(DEV))
    again = _launch.linear_k512(xd, wd)
    torch.cuda.synchronize()
    ref = x.double().numpy() @ w.double().numpy().T
    m = record_parity("linear_k512.y", y.double().cpu().numpy(), ref, str(dtype).replace("torch.", ""))
    assert m["rel_fro"] <= GATE[dtype], m
    assert _rel(yb, ref + b.double().numpy()) <= GATE[dtype]
    assert torch.equal(y, again) and torch.isfinite(y).all()


def test_linear_k512_strided_rows_and_refusals():
    """rows taken from a wider buffer (row stride 1024), and what the kernel does not take"""
    from generative_recommenders_amd.ops import _launch

    g = torch.Generator().manual_seed(3)
    wide = torch.randn(3001, 1024, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(1536, 512, generator=g) / 512**0.5).to(torch.bfloat16).to(DEV)
    x = wide[:, 256:768]
    assert _launch.linear_k512_supported(x, 1536)
    assert torch.equal(_launch.linear_k512(x, w), _launch.linear_k512(x.contiguous(), w))
    assert not _launch.linear_k512_supported(wide[:, 4:516], 1536)                      # 8-byte aligned rows only
    assert not _launch.linear_k512_supported(wide[:, :256], 1536)                       # k != 512
    assert not _launch.linear_k512_supported(x.float(), 1536) and not _launch.linear_k512_supported(x, 1000)
    with pytest.raises(RuntimeError, match="k == 512"):
        _launch.linear_k512(wide[:, :256], w[:, :256].contiguous())


def test_output_stage_dgrad_kernel_against_hipblaslt(monkeypatch):
    """gradients of one STU layer with d y from hstu_linear_k512 (default) against the same layer with torch.mm
    (HSTU_OUT_DGRAD_KERNEL=0): same operands, same single rounding of d y -- summation order at most"""
    from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig

    from generative_recommenders_amd.ops import _launch

    calls = []
    real = _launch.linear_k512
    monkeypatch.setattr(_launch, "linear_k512", lambda *a, **k: (calls.append(1), real(*a, **k))[1])

    def run(env):
        monkeypatch.setenv("HSTU_OUT_DGRAD_KERNEL", env)
        torch.manual_seed(0)
        layer = STULayer(STULayerConfig(embedding_dim=512, num_heads=4, hidden_dim=128, attention_dim=128, output_dropout_ratio=0.0,
                                        use_group_norm=True)).to(DEV)
        g = torch.Generator().manual_seed(1)
        lengths = torch.tensor([200, 13, 0, 177, 200], dtype=torch.int64)
        off = torch.zeros(6, dtype=torch.int64)
        off[1:] = torch.cumsum(lengths, 0)
        x = torch.randn(int(off[-1]), 512, generator=g).to(torch.bfloat16).to(DEV).requires_grad_()
        gy = torch.randn(int(off[-1]), 512, generator=g).to(torch.bfloat16).to(DEV)
        layer(x=x, x_lengths=lengths.to(DEV), x_offsets=off.to(DEV), max_seq_len=200, num_targets=None).backward(gy)
        torch.cuda.synchronize()
        return {"dx": x.grad.clone(), **{n: p.grad.clone() for n, p in layer.named_parameters()}}

    a = run("1")
    assert len(calls) == 1, "the output stage's backward did not go through hstu_linear_k512"
    b = run("0")
    assert len(calls) == 1, "HSTU_OUT_DGRAD_KERNEL=0 still ran the kernel"
    for name in a:
        ra = a[name].double().cpu().numpy()
        rb = b[name].double().cpu().numpy()
        rel = float(np.linalg.norm(ra - rb) / max(np.linalg.norm(rb), 1e-30))
        assert rel <= 3e-3, (name, rel)
    # (at this shape the two are in fact bit-identical: both accumulate k in ascending 16-element MFMA steps in fp32 and round once)
