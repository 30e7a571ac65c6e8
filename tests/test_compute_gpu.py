"""GPU tests of the row kernels and the fused autograd nodes around the projections
(layer norm, SiLU, LN/GroupNorm * u with concat, hstu_compute_uqvk / hstu_compute_output /
hstu_preprocess_and_attention, STULayer / STUStack) against the reference's golden vectors
and the numpy oracle.  fp32 runs use element-wise rtol 1e-3; TF32 does not exist on gfx950
(torch.mm on fp32 is exact fp32), so the fp32 GEMMs are held to the same bar."""

import copy

import numpy as np
import pytest
import torch

from conftest import load_cases, record_parity
from oracle import hstu_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(got, ref, rtol=1e-3, atol_scale=2e-5, what=""):
    g = got.detach().float().cpu().numpy().astype(np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert g.shape == ref.shape, f"{what}: {g.shape} vs {ref.shape}"
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(g - ref)
    m = record_parity(what, g, ref, str(got.dtype).replace("torch.", ""))
    # relative Frobenius gate by output dtype: 1.5 x the largest error measured on MI355X (profiles/r02_parity_errors.md)
    gate = {torch.float32: 1.5e-6, torch.bfloat16: 2.8e-3, torch.float16: 3.2e-4}[got.dtype]
    assert m["rel_fro"] <= gate, f"{what}: relative Frobenius error {m['rel_fro']:.3e} (gate {gate})"
    bad = err > rtol * np.abs(ref) + atol_scale * scale
    assert not bad.any(), f"{what}: {bad.sum()}/{bad.size} out of tolerance, max err {err.max():.3e}, scale {scale:.3e}"


def _t(x, dtype=torch.float32, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV).to(dtype)
    return t.requires_grad_() if grad else t


def _compute(name):
    for c in load_cases("compute.npz"):
        if str(c["name"]) == name:
            return c
    raise KeyError(name)


def test_layer_norm_golden_and_bwd():
    from generative_recommenders_amd.ops.layer_norm import layer_norm

    c = _compute("ln")
    x, w, b = _t(c["x"], grad=True), _t(c["w"], grad=True), _t(c["b"], grad=True)
    y = layer_norm(x, w, b, float(c["eps"]))
    _close(y, c["y"], what="ln y")
    g = torch.randn_like(y)
    y.backward(g)
    dx, dw, db = O.layer_norm_bwd(g.cpu().numpy(), c["x"], c["w"], float(c["eps"]))
    _close(x.grad, dx, what="ln dx")
    _close(w.grad, dw, what="ln dw")
    _close(b.grad, db, what="ln db")


@pytest.mark.parametrize("rows,dim,dtype", [(0, 64, torch.float32), (1, 32, torch.float32), (1000, 512, torch.bfloat16),
                                            (777, 100, torch.float32), (300, 37, torch.bfloat16), (64, 1024, torch.float16),
                                            (5000, 512, torch.float32),
                                            # rows wider than 1024 elements (the reference's kernel takes any D,
                                            # triton_layer_norm.py:340): the wide instance of the row kernels, up to 4096
                                            (200, 2048, torch.bfloat16), (65, 4096, torch.float32), (33, 1536, torch.float16),
                                            (17, 1500, torch.float32)])
def test_layer_norm_sweep(rows, dim, dtype):
    """N in [0, 10000], arbitrary D (ops/tests/layer_norm_test.py:62-80)."""
    from generative_recommenders_amd.ops.layer_norm import layer_norm

    g = torch.Generator().manual_seed(rows + dim)
    x = torch.randn(rows, dim, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(dim, generator=g)).to(dtype)
    b = (0.1 * torch.randn(dim, generator=g)).to(dtype)
    xd, wd, bd = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = layer_norm(xd, wd, bd, 1e-6)
    assert y.shape == (rows, dim)
    if rows == 0:
        return
    ref = O.layer_norm_fwd(x.double().numpy(), w.double().numpy(), b.double().numpy(), 1e-6)
    tol = dict(rtol=1e-3, atol_scale=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol_scale=4e-3)
    _close(y, ref, what="y", **tol)
    gy = torch.randn(rows, dim, generator=g).to(dtype)
    y.backward(gy.to(DEV))
    dx, dw, db = O.layer_norm_bwd(gy.double().numpy(), x.double().numpy(), w.double().numpy(), 1e-6)
    _close(xd.grad, dx, what="dx", **tol)
    wt = dict(rtol=1e-3, atol_scale=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol_scale=1e-2)
    _close(wd.grad, dw, what="dw", **wt)
    _close(bd.grad, db, what="db", **wt)


@pytest.mark.parametrize("heads,hd,group_norm,dtype", [(16, 128, True, torch.bfloat16), (16, 128, False, torch.float32),
                                                       (8, 512, True, torch.float32), (12, 100, False, torch.bfloat16)])
def test_norm_mul_wide_rows(heads, hd, group_norm, dtype):
    """H x hidden_dim = 1200 .. 4096 (HSTU-large: 16 heads of 128): y = [u, attn, u * Norm(attn)] and its backward against the
    oracle.  Past 4096 elements per row the op says so."""
    from generative_recommenders_amd.ops import _launch

    rows, dim = 77, heads * hd
    g = torch.Generator().manual_seed(dim)
    attn = torch.randn(rows, dim, generator=g).to(dtype)
    u = torch.randn(rows, dim, generator=g).to(dtype)
    width = heads if group_norm else dim
    w = (1 + 0.1 * torch.randn(width, generator=g)).to(dtype)
    b = (0.1 * torch.randn(width, generator=g)).to(dtype)
    y, mean, rstd = _launch.norm_mul_fwd(attn.to(DEV), u.to(DEV), w.to(DEV), b.to(DEV), 1e-5, heads, hd, group_norm, True)
    ref = O.norm_mul(attn.double().numpy(), u.double().numpy(), w.double().numpy(), b.double().numpy(), 1e-5, True, group_norm, heads, hd)
    tol = dict(rtol=1e-3, atol_scale=2e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol_scale=4e-3)
    _close(y, ref, what="wide norm_mul y", **tol)
    # backward: against torch autograd on the same formula in fp64 on the CPU
    a64, u64 = attn.double().requires_grad_(), u.double().requires_grad_()
    w64, b64 = w.double().requires_grad_(), b.double().requires_grad_()
    if group_norm:
        xh = a64.view(rows, heads, hd)
        n = (xh - xh.mean(-1, keepdim=True)) / torch.sqrt(xh.var(-1, unbiased=False, keepdim=True) + 1e-5)
        n = (n * w64.view(1, heads, 1) + b64.view(1, heads, 1)).reshape(rows, dim)
    else:
        n = torch.nn.functional.layer_norm(a64, (dim,), w64, b64, 1e-5)
    y64 = torch.cat([u64, a64, u64 * n], dim=1)
    dy = torch.randn(rows, 3 * dim, generator=g).to(dtype)
    y64.backward(dy.double())
    dattn, du, dw, db = _launch.norm_mul_bwd(dy.to(DEV), attn.to(DEV), u.to(DEV), w.to(DEV), b.to(DEV), mean, rstd, heads, hd,
                                             group_norm, True)
    _close(dattn, a64.grad.numpy(), what="wide norm_mul dattn", **tol)
    _close(du, u64.grad.numpy(), what="wide norm_mul du", **tol)
    wt = dict(rtol=1e-3, atol_scale=1e-4)
    assert dw.dtype == torch.float32
    _close(dw, w64.grad.numpy(), what="wide norm_mul dw", **wt)
    _close(db, b64.grad.numpy(), what="wide norm_mul db", **wt)


def test_rows_past_4096_are_refused():
    from generative_recommenders_amd.ops.layer_norm import layer_norm

    x = torch.randn(4, 4104, device=DEV)
    with pytest.raises(RuntimeError, match="exceeds"):
        layer_norm(x, torch.ones(4104, device=DEV), torch.zeros(4104, device=DEV), 1e-5)


def test_uvqk_golden_fwd_bwd():
    from generative_recommenders_amd.ops.hstu_compute import hstu_compute_uqvk

    c = _compute("uvqk")
    x, nw, nb = _t(c["x"], grad=True), _t(c["nw"], grad=True), _t(c["nb"], grad=True)
    W, beta = _t(c["W"], grad=True), _t(c["beta"], grad=True)
    u, q, k, v = hstu_compute_uqvk(x, nw, nb, 1e-6, int(c["H"]), int(c["A"]), int(c["Hd"]), W, beta)
    for got, name in ((u, "u"), (q, "q"), (k, "k"), (v, "v")):
        _close(got, c[name], what=name)
    loss = (u * _t(c["gu"])).sum() + (q * _t(c["gq"])).sum() + (k * _t(c["gk"])).sum() + (v * _t(c["gv"])).sum()
    loss.backward()
    _close(x.grad, c["dx"], what="dx", atol_scale=1e-4)
    _close(nw.grad, c["dnw"], what="dnw", atol_scale=1e-4)
    _close(nb.grad, c["dnb"], what="dnb", atol_scale=1e-4)
    _close(W.grad, c["dW"], what="dW", atol_scale=1e-4)
    _close(beta.grad, c["dbeta"], what="dbeta", atol_scale=1e-4)


@pytest.mark.parametrize("name", ["out_ln", "out_ln_cat", "out_gn_cat"])
@pytest.mark.parametrize("recompute_y", [False, True])
def test_compute_output_golden_fwd_bwd(name, recompute_y):
    from generative_recommenders_amd.ops.hstu_compute import hstu_compute_output

    c = _compute(name)
    attn, u, x = _t(c["attn"], grad=True), _t(c["u"], grad=True), _t(c["x"], grad=True)
    nw, nb, Wo = _t(c["nw"], grad=True), _t(c["nb"], grad=True), _t(c["Wo"], grad=True)
    y = hstu_compute_output(attn=attn, u=u, x=x, norm_weight=nw, norm_bias=nb, norm_eps=1e-6, output_weight=Wo,
                            num_heads=int(c["H"]), linear_dim=int(c["Ld"]), dropout_ratio=0.0, training=False,
                            concat_ux=bool(c["cat"]), group_norm=bool(c["gn"]), recompute_y_in_backward=recompute_y)
    _close(y, c["y"], what="y")
    y.backward(_t(c["gy"]))
    for got, name2 in ((attn.grad, "dattn"), (u.grad, "du"), (x.grad, "dx"), (nw.grad, "dnw"), (nb.grad, "dnb"),
                       (Wo.grad, "dWo")):
        _close(got, c[name2], what=name2, atol_scale=1e-4)


def test_compute_output_shape_of_reference_test_bf16():
    """N=1000 rows, H=4, linear_dim=128, D=128, group norm + concat (ops/tests/hstu_compute_test.py:36-66)."""
    from generative_recommenders_amd.ops.hstu_compute import hstu_compute_output

    g = torch.Generator().manual_seed(3)
    N, H, Ld, D = 1000, 4, 128, 128
    attn = torch.randn(N, H * Ld, generator=g).bfloat16()
    u = torch.randn(N, H * Ld, generator=g).bfloat16()
    x = torch.randn(N, D, generator=g).bfloat16()
    nw = (1 + 0.1 * torch.randn(H, generator=g)).bfloat16()
    nb = (0.1 * torch.randn(H, generator=g)).bfloat16()
    Wo = (0.05 * torch.randn(3 * H * Ld, D, generator=g)).bfloat16()
    y = hstu_compute_output(attn=attn.to(DEV), u=u.to(DEV), x=x.to(DEV), norm_weight=nw.to(DEV), norm_bias=nb.to(DEV),
                            norm_eps=1e-6, output_weight=Wo.to(DEV), num_heads=H, linear_dim=Ld, dropout_ratio=0.0,
                            training=False, concat_ux=True, group_norm=True, recompute_y_in_backward=True)
    ref = O.hstu_compute_output(attn.double().numpy(), u.double().numpy(), x.double().numpy(), nw.double().numpy(),
                                nb.double().numpy(), 1e-6, Wo.double().numpy(), H, Ld, True, True)
    _close(y, ref, rtol=3e-2, atol_scale=1e-2, what="bf16 output")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("heads,hd,group_norm", [(4, 128, True), (4, 64, False), (3, 40, True), (2, 24, False), (8, 256, True)])
@pytest.mark.parametrize("concat", [False, True])
def test_norm_mul_silu_matches_separate_kernels(dtype, heads, hd, group_norm, concat):
    """hstu_norm_mul_silu_fwd / _bwd (ABI v7: u read in place, as the pre-activation slice of a wider buffer, SiLU and SiLU'
    applied inside) against hstu_silu_fwd -> hstu_norm_mul_dropout_fwd and hstu_norm_mul_dropout_bwd -> hstu_silu_bwd on
    the same inputs, with dropout: every kernel variant (group-norm fast path single / multi chunk, generic vector and
    scalar paths, the wide instance), rows that are not 16-byte multiples (40, 24 elements per head: scalar path)."""
    from generative_recommenders_amd.ops import _launch

    rows, dim = 333, heads * hd
    g = torch.Generator().manual_seed(5)
    buf = torch.randn(rows, 4 * dim + 8, generator=g).to(DEV).to(dtype)       # u = columns [8, 8 + dim) of a wider buffer
    u_pre = buf[:, 8:8 + dim]
    attn = torch.randn(rows, dim, generator=g).to(DEV).to(dtype)
    width = heads if group_norm else dim
    w = (1 + 0.1 * torch.randn(width, generator=g)).to(DEV).to(dtype)
    b = (0.1 * torch.randn(width, generator=g)).to(DEV).to(dtype)
    dy = torch.randn(rows, 3 * dim if concat else dim, generator=g).to(DEV).to(dtype)
    p, seed = 0.25, 0x1234567
    # separate passes
    u = _launch.silu_fwd(u_pre)
    y0, m0, r0 = _launch.norm_mul_fwd(attn, u, w, b, 1e-6, heads, hd, group_norm, concat, p, seed)
    dattn0, du0, dw0, db0 = _launch.norm_mul_bwd(dy, attn, u, w, b, m0, r0, heads, hd, group_norm, concat, p, seed)
    dpre0 = _launch.silu_bwd(du0, u_pre)
    # fused
    y1, m1, r1 = _launch.norm_mul_fwd(attn, u_pre, w, b, 1e-6, heads, hd, group_norm, concat, p, seed, u_is_preactivation=True)
    dbuf = torch.full_like(buf, 7.0)
    dattn1, dpre1, dw1, db1 = _launch.norm_mul_bwd(dy, attn, u_pre, w, b, m1, r1, heads, hd, group_norm, concat, p, seed,
                                                   u_is_preactivation=True, du=dbuf[:, 8:8 + dim])
    exact = dtype != torch.float32          # fp32: an fma may be contracted differently in the fused kernel
    for name, a, c in (("y", y1, y0), ("mean", m1, m0), ("rstd", r1, r0), ("dattn", dattn1, dattn0), ("d u_pre", dpre1, dpre0)):
        if exact:
            assert torch.equal(a, c), f"{name}: max abs diff {(a.float() - c.float()).abs().max().item():.3e}"
        else:
            torch.testing.assert_close(a, c, rtol=2e-6, atol=2e-6 * float(c.abs().max()), msg=lambda m_: f"{name}: {m_}")
    torch.testing.assert_close(dw1, dw0, rtol=1e-5, atol=1e-5 * float(dw0.abs().max()))
    torch.testing.assert_close(db1, db0, rtol=1e-5, atol=1e-5 * float(db0.abs().max()))
    assert dpre1.data_ptr() == dbuf[:, 8:8 + dim].data_ptr()                       # written in place ...
    assert bool((dbuf[:, :8] == 7.0).all()) and bool((dbuf[:, 8 + dim:] == 7.0).all())   # ... and nothing around it


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,dim", [(257, 512), (64, 100), (33, 2048)])
def test_layer_norm_bwd_residual(rows, dim, dtype):
    """hstu_layer_norm_bwd_residual: dx = LayerNorm'(dy) + dresidual in one pass == the two-pass result, rounded the same way."""
    from generative_recommenders_amd.ops import _launch

    g = torch.Generator().manual_seed(9)
    x = torch.randn(rows, dim, generator=g).to(DEV).to(dtype)
    dy = torch.randn(rows, dim, generator=g).to(DEV).to(dtype)
    dres = torch.randn(rows, dim, generator=g).to(DEV).to(dtype)
    w = (1 + 0.1 * torch.randn(dim, generator=g)).to(DEV).to(dtype)
    b = torch.zeros(dim, device=DEV, dtype=dtype)
    _, mean, rstd = _launch.layer_norm_fwd(x, w, b, 1e-6)
    dx0, dw0, db0 = _launch.layer_norm_bwd(dy, x, w, mean, rstd)
    dx1, dw1, db1 = _launch.layer_norm_bwd(dy, x, w, mean, rstd, dresidual=dres)
    assert torch.equal(dx1, dx0 + dres)
    assert torch.equal(dw1, dw0) and torch.equal(db1, db0)


def _load_stack(c, dtype=torch.float32, **cfg_over):
    from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig, STUStack

    D, H, A, Hd = int(c["D"]), int(c["H"]), int(c["A"]), int(c["Hd"])
    layers = []
    for gn in (False, True):
        cfg = dict(embedding_dim=D, num_heads=H, hidden_dim=Hd, attention_dim=A, output_dropout_ratio=0.0,
                   causal=True, target_aware=True, max_attn_len=None, attn_alpha=None, use_group_norm=gn,
                   recompute_normed_x=True, recompute_uvqk=True, recompute_y=True, sort_by_length=True,
                   contextual_seq_len=0)
        cfg.update(cfg_over)
        layers.append(STULayer(STULayerConfig(**cfg)))
    stack = STUStack(layers)
    sd = {k[2:]: torch.from_numpy(v) for k, v in c.items() if k.startswith("p:")}
    stack.load_state_dict(sd)  # reference parameter names must load unchanged
    return stack.to(DEV).to(dtype)


@pytest.mark.parametrize("flags", [(True, True, True), (False, False, False), (True, False, True), (False, True, False)])
def test_stu_stack_golden_fwd_bwd(flags):
    """2-layer STUStack (LayerNorm layer + GroupNorm layer) vs the reference's PYTORCH-kernel
    stack on identical parameters and inputs; all recompute-flag combinations must agree
    (modules/tests/stu_test.py:48-72)."""
    c = load_cases("stu.npz")[0]
    stack = _load_stack(c, recompute_normed_x=flags[0], recompute_uvqk=flags[1], recompute_y=flags[2])
    x = _t(c["x"], grad=True)
    y = stack(x=x, x_lengths=torch.from_numpy(c["lengths"]).to(DEV), x_offsets=torch.from_numpy(c["offsets"]).to(DEV),
              max_seq_len=int(c["N"]), num_targets=torch.from_numpy(c["num_targets"]).to(DEV))
    _close(y, c["y"], what="stack y", atol_scale=1e-4)
    y.backward(_t(c["gy"]))
    _close(x.grad, c["dx"], what="stack dx", atol_scale=2e-4)
    for name, p in stack.named_parameters():
        _close(p.grad, c["g:" + name], what="grad " + name, rtol=2e-3, atol_scale=3e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("group_norm", [False, True])
@pytest.mark.parametrize("recompute", [True, False])
@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_fused_layer_node_matches_two_nodes(dtype, group_norm, recompute, dropout):
    """STULayer.forward as ONE autograd node (SiLU folded into the output-stage kernels, the residual's gradient into the
    layer-norm backward) against the reference-shaped pair of nodes: the same kernels' arithmetic rounded at the same
    places, so the output and every gradient must agree bit for bit (16-bit activations; to an ulp or two in fp32) -- fp32 master parameters next to `dtype` activations,
    targets, sort_by_length, with the fused dropout (same seed) and all recompute flags on or off."""
    from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig

    D, H, Hd, A, N, B = 256, 4, 64, 64, 96, 7
    g = torch.Generator().manual_seed(11)
    lengths = torch.randint(1, N + 1, (B,), generator=g)
    off = torch.zeros(B + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lengths, 0)
    nt = torch.minimum(torch.randint(0, 9, (B,), generator=g), lengths)
    x0 = torch.randn(int(off[-1]), D, generator=g)
    gy = torch.randn(int(off[-1]), D, generator=g)
    torch.manual_seed(3)
    layer = STULayer(STULayerConfig(embedding_dim=D, num_heads=H, hidden_dim=Hd, attention_dim=A, output_dropout_ratio=dropout,
                                    causal=True, target_aware=True, max_attn_len=None, attn_alpha=None, use_group_norm=group_norm,
                                    recompute_normed_x=recompute, recompute_uvqk=recompute, recompute_y=recompute,
                                    sort_by_length=True, contextual_seq_len=0)).to(DEV).train()
    with torch.no_grad():
        for p in layer.parameters():      # the norm weights start at exactly 1 / 0: make every gradient path non-trivial
            p.add_(0.05 * torch.randn(p.shape, generator=g).to(DEV))
    res = {}
    for fused in (True, False):
        layer.fuse_layer = fused
        layer.zero_grad(set_to_none=True)
        x = x0.to(DEV).to(dtype).requires_grad_()
        torch.manual_seed(17)             # the dropout seed is drawn from torch's CPU generator
        y = layer(x=x, x_lengths=lengths.to(DEV), x_offsets=off.to(DEV), max_seq_len=N, num_targets=nt.to(DEV))
        y.backward(gy.to(DEV).to(dtype))
        res[fused] = [y.detach(), x.grad] + [p.grad for p in layer.parameters()]
    names = ["y", "dx"] + [n for n, _ in layer.named_parameters()]
    for n, a, b in zip(names, res[True], res[False]):
        assert a.dtype == b.dtype and a.shape == b.shape, n
        if dtype == torch.float32:      # (fp32 rows: an fma contracted differently here or there -- an ulp of u, carried through the GEMMs)
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()), msg=lambda m: f"{n}: {m}")
        else:
            assert torch.equal(a, b), f"{n}: fused node differs from the two-node path (max abs diff {(a.float() - b.float()).abs().max().item():.3e})"
    if dropout > 0:
        assert (res[True][0] != 0).any()


def test_stu_cached_forward_equals_full_forward():
    """prefill + cached_forward == the delta rows of a full forward
    (modules/tests/stu_test.py:341-457: every delta row is a target, same max_seq_len in
    both passes so the 1/N scale agrees)."""
    from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig, STUStack
    from generative_recommenders_amd.ops.jagged_tensors import asynchronous_complete_cumsum, split_2D_jagged

    torch.manual_seed(0)
    D, H, A, Hd, delta = 64, 2, 32, 32, 20
    for ctx in (0, 4):
        layers = [STULayer(STULayerConfig(embedding_dim=D, num_heads=H, hidden_dim=Hd, attention_dim=A,
                                          output_dropout_ratio=0.0, target_aware=True, use_group_norm=bool(i),
                                          contextual_seq_len=ctx), is_inference=True) for i in range(2)]
        stack = STUStack(layers, is_inference=True).to(DEV).eval()
        B, max_uih = 5, 60
        g = torch.Generator().manual_seed(1)
        nt = torch.randint(delta, 2 * delta + 1, (B,), generator=g)
        lengths = torch.randint(1, max_uih + 1, (B,), generator=g) + nt + ctx
        N = max_uih + 2 * delta + ctx
        x = torch.randn(int(lengths.sum()), D, generator=g).to(DEV)
        lend, ntd = lengths.to(DEV), nt.to(DEV)
        offd = asynchronous_complete_cumsum(lend)
        with torch.no_grad():
            full = stack(x=x, x_lengths=lend, x_offsets=offd, max_seq_len=N, num_targets=ntd)
            prime_len = lend - delta
            prime_off = asynchronous_complete_cumsum(prime_len)
            _, full_tail = split_2D_jagged(N, full, None, None, None, delta, prime_off, None)
            prime_x, delta_x = split_2D_jagged(N, x, None, None, None, delta, prime_off, None)
            stack(x=prime_x, x_lengths=prime_len, x_offsets=prime_off, max_seq_len=N, num_targets=ntd - delta,
                  max_kv_caching_len=N - delta, kv_caching_lengths=prime_len)
            inc = stack.cached_forward(delta_x=delta_x, num_targets=ntd)
        torch.testing.assert_close(inc, full_tail, rtol=1e-4, atol=1e-5)


def test_stu_cached_forward_in_place_append_matches_rebuild():
    """M-FALCON microbatching: several cached_forward calls on one cache.  Under no_grad the [cache ; delta] buffers
    are kept and only the delta rows are rewritten (hstu_jagged_write_tail); every microbatch must give exactly what
    the reference's rebuild-by-concat gives (here: the same layer run with grad mode on, which takes the concat path),
    the cache itself must be untouched, and a changed microbatch size or cache must drop the buffers."""
    from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig
    from generative_recommenders_amd.ops.jagged_tensors import asynchronous_complete_cumsum

    torch.manual_seed(3)
    D, H, A, Hd = 64, 2, 32, 32
    layer = STULayer(STULayerConfig(embedding_dim=D, num_heads=H, hidden_dim=Hd, attention_dim=A, output_dropout_ratio=0.0,
                                    target_aware=True, use_group_norm=True), is_inference=True).to(DEV).eval()
    B, N = 6, 96
    g = torch.Generator().manual_seed(2)
    lengths = torch.randint(1, 70, (B,), generator=g).to(DEV)
    off = asynchronous_complete_cumsum(lengths)
    x = torch.randn(int(lengths.sum()), D, generator=g).to(DEV)
    with torch.no_grad():
        layer(x=x, x_lengths=lengths, x_offsets=off, max_seq_len=N, num_targets=torch.zeros_like(lengths),
              max_kv_caching_len=N, kv_caching_lengths=lengths)
    k0, v0 = layer.k_cache.clone(), layer.v_cache.clone()
    for step, delta in enumerate((16, 16, 16, 8, 8)):
        dx = torch.randn(B * delta, D, generator=g).to(DEV)
        nt = torch.full((B,), delta, device=DEV, dtype=lengths.dtype)
        with torch.no_grad():
            fast = layer.cached_forward(delta_x=dx, num_targets=nt).clone()
            assert layer._kv_full is not None and layer._kv_full[2][0] == delta
            kept = layer._kv_full[0].data_ptr()
        with torch.enable_grad():                       # concat path; leaves the persistent buffers alone
            slow = layer.cached_forward(delta_x=dx, num_targets=nt).detach()
        assert torch.equal(fast, slow), f"microbatch {step}"
        assert layer._kv_full[0].data_ptr() == kept
        assert torch.equal(layer.k_cache, k0) and torch.equal(layer.v_cache, v0)
    # re-priming the cache drops the buffers
    with torch.no_grad():
        layer(x=x, x_lengths=lengths, x_offsets=off, max_seq_len=N, num_targets=torch.zeros_like(lengths),
              max_kv_caching_len=N, kv_caching_lengths=lengths)
    assert layer._kv_full is None


def test_jagged_write_tail_bit_exact():
    from generative_recommenders_amd.ops import _launch
    rng = np.random.default_rng(0)
    for dtype, dim in ((torch.bfloat16, 64), (torch.float32, 7), (torch.int64, 1), (torch.uint8, 3)):
        lengths = np.array([5, 0, 9, 3, 3], dtype=np.int64) + 3           # every region holds at least `tail` rows
        tail = 3
        off = np.zeros(6, dtype=np.int64); off[1:] = np.cumsum(lengths)
        base = torch.from_numpy(rng.integers(0, 100, (int(off[-1]), dim))).to(dtype).to(DEV)
        new = torch.from_numpy(rng.integers(100, 200, (5 * tail, dim))).to(dtype).to(DEV)
        want = base.clone()
        for b in range(5):
            want[off[b + 1] - tail: off[b + 1]] = new[b * tail: (b + 1) * tail]
        got = _launch.jagged_write_tail_(base, new, torch.from_numpy(off).to(DEV), tail)
        assert got.data_ptr() == base.data_ptr() and torch.equal(got, want)
    with pytest.raises(RuntimeError):
        _launch.jagged_write_tail_(base, new[:, :1].float(), torch.from_numpy(off).to(DEV), tail)


def test_silu_matches_torch():
    from generative_recommenders_amd.ops import _launch

    x = torch.randn(100, 96, device=DEV)
    sl = x[:, 16:80]
    y = _launch.silu_fwd(sl)
    torch.testing.assert_close(y, torch.nn.functional.silu(sl), rtol=1e-5, atol=1e-6)
    g = torch.randn_like(y)
    ref = torch.autograd.grad(torch.nn.functional.silu(sl.clone().requires_grad_()), [], allow_unused=True) if False else None
    del ref
    s = sl.clone().requires_grad_()
    torch.nn.functional.silu(s).backward(g)
    torch.testing.assert_close(_launch.silu_bwd(g, sl), s.grad, rtol=1e-5, atol=1e-6)
