#!/usr/bin/env python3
"""Mint golden fixtures from the REFERENCE's own PyTorch path.

Run in the build container only (needs /root/reference, which does not exist on
the GPU box):

    python tests/golden/make_golden.py

It imports ``generative_recommenders`` from /root/reference unmodified, supplies
the three absent ``fbgemm_gpu`` ops through ``_fbgemm_shim`` (pure-torch
CompositeImplicitAutograd restatements), runs the HammerKernel.PYTORCH branch of
each hot-path function on seeded CPU inputs drawn from the distributions of the
reference's tests (ops/tests/hstu_attention_test.py:62-120) and stores inputs +
outputs + gradients as small ``.npz`` files next to this script.  The committed
``.npz`` files are what ``tests/`` reads; this script is committed so they can
be regenerated.
"""

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)

import numpy as np
import torch

import _fbgemm_shim  # noqa: F401  (registers torch.ops.fbgemm.*)

from generative_recommenders.common import HammerKernel  # noqa: E402
from generative_recommenders.ops.hstu_attention import delta_hstu_mha, hstu_mha  # noqa: E402
from generative_recommenders.ops.hstu_compute import (  # noqa: E402
    hstu_compute_output,
    hstu_compute_uqvk,
)
from generative_recommenders.ops.jagged_tensors import (  # noqa: E402
    concat_2D_jagged,
    hstu_concat_l2_embeddings,
    hstu_split_l2_embeddings,
    split_2D_jagged,
)
from generative_recommenders.ops.layer_norm import SwishLayerNorm, layer_norm, swish_layer_norm  # noqa: E402
from generative_recommenders.modules.stu import STULayer, STULayerConfig, STUStack  # noqa: E402

PT = HammerKernel.PYTORCH


def _np(t):
    return t.detach().cpu().numpy()


def _lengths(gen, B, max_uih_len, max_targets, contextual, with_targets):
    """lengths = randint(max_uih_len + 1) + num_targets + contextual
    (ops/tests/hstu_attention_test.py:62-85)."""
    uih = torch.randint(0, max_uih_len + 1, (B,), generator=gen)
    if with_targets:
        nt = torch.randint(1, max_targets + 1, (B,), generator=gen)
    else:
        nt = torch.zeros(B, dtype=torch.int64)
    lengths = uih + nt + contextual
    return lengths, nt


def attention_cases():
    gen = torch.Generator().manual_seed(20250629)
    cases = []
    cfgs = [
        # B, H, max_uih, max_tgt, dqk, dv, targets, window, contextual, min_full
        (4, 2, 40, 6, 16, 32, False, False, 0, 0),
        (5, 2, 40, 6, 32, 16, True, False, 0, 0),
        (4, 1, 50, 8, 16, 16, False, True, 0, 0),
        (6, 2, 50, 8, 32, 32, True, True, 0, 0),
        (4, 2, 40, 6, 16, 32, False, False, 5, 0),
        (5, 3, 40, 6, 16, 16, True, True, 5, 0),
        (4, 2, 60, 6, 16, 16, True, True, 0, 7),
        (4, 2, 60, 6, 32, 16, True, True, 4, 9),
        (3, 2, 48, 4, 64, 64, True, False, 0, 0),
    ]
    for ci, (B, H, mu, mt, dqk, dv, tg, win, ctx, mf) in enumerate(cfgs):
        lengths, nt = _lengths(gen, B, mu, mt, ctx, tg)
        if ci == 1:
            lengths[0] = nt[0] + ctx  # zero-history user
        N = int(lengths.max().item())
        N = max(N, 1)
        offsets = torch.zeros(B + 1, dtype=torch.int64)
        offsets[1:] = torch.cumsum(lengths, 0)
        Ltot = int(offsets[-1])
        w = int(torch.randint(1, max(mu // 5, 2), (1,), generator=gen)) if win else 0
        alpha = 1.0 / (dqk**0.5)
        q = torch.empty(Ltot, H, dqk).uniform_(-0.1, 0.1, generator=gen).requires_grad_()
        k = torch.empty(Ltot, H, dqk).uniform_(-0.1, 0.1, generator=gen).requires_grad_()
        v = torch.empty(Ltot, H, dv).uniform_(-0.1, 0.1, generator=gen).requires_grad_()
        dout = torch.randn(Ltot, H, dv, generator=gen) * 0.1
        out = hstu_mha(
            max_seq_len=N, alpha=alpha, q=q, k=k, v=v, seq_offsets=offsets, causal=True,
            dropout_pr=0.0, training=False, num_targets=nt if tg else None, max_attn_len=w,
            contextual_seq_len=ctx, min_full_attn_seq_len=mf, kernel=PT,
        )
        out.backward(dout)
        cases.append(dict(
            N=N, alpha=alpha, H=H, dqk=dqk, dv=dv, offsets=_np(offsets),
            num_targets=_np(nt) if tg else None, max_attn_len=w, contextual=ctx, min_full=mf,
            q=_np(q), k=_np(k), v=_np(v), dout=_np(dout), out=_np(out),
            dq=_np(q.grad), dk=_np(k.grad), dv_=_np(v.grad),
        ))
    return cases


def delta_cases():
    gen = torch.Generator().manual_seed(77)
    cases = []
    for (B, H, mu, delta, dqk, dv, tg, win, ctx) in [
        (4, 2, 40, 6, 16, 32, True, False, 0),
        (3, 2, 50, 8, 32, 32, True, True, 4),
        (4, 1, 30, 5, 16, 16, False, False, 0),
    ]:
        uih = torch.randint(0, mu + 1, (B,), generator=gen)
        nt = torch.randint(1, delta + 1, (B,), generator=gen)
        lengths = uih + delta + ctx
        N = int(lengths.max())
        offsets = torch.zeros(B + 1, dtype=torch.int64)
        offsets[1:] = torch.cumsum(lengths, 0)
        Ltot = int(offsets[-1])
        w = int(torch.randint(1, max(mu // 5, 2), (1,), generator=gen)) if win else 0
        alpha = 1.0 / (dqk**0.5)
        dq_ = torch.empty(B * delta, H, dqk).uniform_(-0.1, 0.1, generator=gen)
        k = torch.empty(Ltot, H, dqk).uniform_(-0.1, 0.1, generator=gen)
        v = torch.empty(Ltot, H, dv).uniform_(-0.1, 0.1, generator=gen)
        out = delta_hstu_mha(
            max_seq_len=N, alpha=alpha, delta_q=dq_, k=k, v=v, seq_offsets=offsets,
            num_targets=nt if tg else None, max_attn_len=w, contextual_seq_len=ctx, kernel=PT,
        )
        cases.append(dict(
            N=N, alpha=alpha, delta=delta, offsets=_np(offsets), num_targets=_np(nt) if tg else None,
            max_attn_len=w, contextual=ctx, delta_q=_np(dq_), k=_np(k), v=_np(v), out=_np(out),
        ))
    return cases


def jagged_cases():
    gen = torch.Generator().manual_seed(5)
    cases = []
    for (B, ma, mb, D, dense_a, dense_b) in [
        (4, 20, 30, 13, False, False),
        (5, 25, 10, 24, True, False),
        (3, 20, 15, 10, False, True),
        (6, 8, 8, 30, False, False),
    ]:
        la = torch.full((B,), ma) if dense_a else torch.randint(0, ma + 1, (B,), generator=gen)
        lb = torch.full((B,), mb) if dense_b else torch.randint(0, mb + 1, (B,), generator=gen)
        oa = torch.zeros(B + 1, dtype=torch.int64); oa[1:] = torch.cumsum(la, 0)
        ob = torch.zeros(B + 1, dtype=torch.int64); ob[1:] = torch.cumsum(lb, 0)
        va = torch.randn(int(oa[-1]), D, generator=gen)
        vb = torch.randn(int(ob[-1]), D, generator=gen)
        cat = concat_2D_jagged(
            max_seq_len=ma + mb, values_left=va, values_right=vb, max_len_left=ma, max_len_right=mb,
            offsets_left=None if dense_a else oa, offsets_right=None if dense_b else ob, kernel=PT,
        )
        sl, sr = split_2D_jagged(
            max_seq_len=ma + mb, values=cat, max_len_left=ma if dense_a else None,
            max_len_right=mb if dense_b else None, offsets_left=None if dense_a else oa,
            offsets_right=None if dense_b else ob, kernel=PT,
        )
        cases.append(dict(ma=ma, mb=mb, dense_a=dense_a, dense_b=dense_b, oa=_np(oa), ob=_np(ob),
                          va=_np(va), vb=_np(vb), cat=_np(cat), split_l=_np(sl), split_r=_np(sr)))
    # l2-embedding prefix variants
    B, mp, ml, D, ctx = 4, 12, 16, 8, 3
    lp = torch.randint(0, mp + 1, (B,), generator=gen)
    ll = torch.randint(ctx, ml + 1, (B,), generator=gen)
    op = torch.zeros(B + 1, dtype=torch.int64); op[1:] = torch.cumsum(lp, 0)
    ol = torch.zeros(B + 1, dtype=torch.int64); ol[1:] = torch.cumsum(ll, 0)
    px = torch.randn(int(op[-1]), D, generator=gen)
    lx = torch.randn(int(ol[-1]), D, generator=gen)
    cat = hstu_concat_l2_embeddings(max_prefix_len=mp, prefix_x=px, prefix_offsets=op, max_l2_len=ml,
                                    l2_x=lx, l2_offsets=ol, contextual_seq_len=ctx, kernel=PT)
    sp, sl2 = hstu_split_l2_embeddings(max_seq_len=mp + ml, x=cat, prefix_offsets=op, l2_offsets=ol,
                                       contextual_seq_len=ctx, kernel=PT)
    l2case = dict(mp=mp, ml=ml, ctx=ctx, op=_np(op), ol=_np(ol), px=_np(px), lx=_np(lx),
                  cat=_np(cat), split_p=_np(sp), split_l=_np(sl2))
    return cases, l2case


def compute_cases():
    gen = torch.Generator().manual_seed(9)
    out = {}
    # layer norm
    x = torch.randn(37, 48, generator=gen)
    w = torch.randn(48, generator=gen); b = torch.randn(48, generator=gen)
    out["ln"] = dict(x=_np(x), w=_np(w), b=_np(b), eps=1e-6, y=_np(layer_norm(x, w, b, 1e-6, kernel=PT)))
    # uvqk
    N, D, H, A, Hd = 29, 32, 2, 16, 24
    x = torch.randn(N, D, generator=gen).requires_grad_()
    nw = (1 + 0.1 * torch.randn(D, generator=gen)).requires_grad_()
    nb = (0.1 * torch.randn(D, generator=gen)).requires_grad_()
    W = (0.1 * torch.randn(D, 2 * H * (A + Hd), generator=gen)).requires_grad_()
    beta = (0.1 * torch.randn(2 * H * (A + Hd), generator=gen)).requires_grad_()
    u, q, k, v = hstu_compute_uqvk(x, nw, nb, 1e-6, H, A, Hd, W, beta, kernel=PT)
    g = [torch.randn(t.shape, generator=gen) for t in (u, q, k, v)]
    (u * g[0]).sum().add((q * g[1]).sum()).add((k * g[2]).sum()).add((v * g[3]).sum()).backward()
    out["uvqk"] = dict(x=_np(x), nw=_np(nw), nb=_np(nb), W=_np(W), beta=_np(beta), H=H, A=A, Hd=Hd,
                       u=_np(u), q=_np(q), k=_np(k), v=_np(v), gu=_np(g[0]), gq=_np(g[1]), gk=_np(g[2]),
                       gv=_np(g[3]), dx=_np(x.grad), dnw=_np(nw.grad), dnb=_np(nb.grad), dW=_np(W.grad),
                       dbeta=_np(beta.grad))
    # output op: LN / GN x concat
    for name, gn, cat in [("out_ln", False, False), ("out_ln_cat", False, True), ("out_gn_cat", True, True)]:
        N, H, Ld, D = 23, 2, 16, 24
        attn = torch.randn(N, H * Ld, generator=gen).requires_grad_()
        u = torch.randn(N, H * Ld, generator=gen).requires_grad_()
        x = torch.randn(N, D, generator=gen).requires_grad_()
        nshape = H if gn else H * Ld
        nw = (1 + 0.1 * torch.randn(nshape, generator=gen)).requires_grad_()
        nb = (0.1 * torch.randn(nshape, generator=gen)).requires_grad_()
        Wo = (0.1 * torch.randn((3 if cat else 1) * H * Ld, D, generator=gen)).requires_grad_()
        y = hstu_compute_output(attn=attn, u=u, x=x, norm_weight=nw, norm_bias=nb, norm_eps=1e-6,
                                output_weight=Wo, num_heads=H, linear_dim=Ld, dropout_ratio=0.0,
                                training=False, concat_ux=cat, group_norm=gn,
                                recompute_y_in_backward=False, kernel=PT)
        gy = torch.randn(y.shape, generator=gen)
        y.backward(gy)
        out[name] = dict(attn=_np(attn), u=_np(u), x=_np(x), nw=_np(nw), nb=_np(nb), Wo=_np(Wo), H=H, Ld=Ld,
                         gn=gn, cat=cat, y=_np(y), gy=_np(gy), dattn=_np(attn.grad), du=_np(u.grad),
                         dx=_np(x.grad), dnw=_np(nw.grad), dnb=_np(nb.grad), dWo=_np(Wo.grad))
    return out


def swish_layer_norm_cases():
    """swish_layer_norm(kernel=PYTORCH) (ops/layer_norm.py:79-112 -> ops/pytorch/pt_layer_norm.py:41-62) forward + autograd, and the
    SwishLayerNorm module at the width DlrmHSTU uses it (modules/dlrm_hstu.py:144)"""
    gen = torch.Generator().manual_seed(77)
    cases = []
    for rows, dim, shift in [(37, 48, 0.0), (5, 512, 0.0), (64, 200, 3.0), (1, 8, 0.0), (33, 1000, 0.5)]:
        x = (torch.randn(rows, dim, generator=gen) * (0.5 + torch.rand(rows, 1, generator=gen)) + shift).requires_grad_()
        w = (1 + 0.2 * torch.randn(dim, generator=gen)).requires_grad_()
        b = (0.2 * torch.randn(dim, generator=gen)).requires_grad_()
        y = swish_layer_norm(x, w, b, 1e-5, kernel=PT)
        gy = torch.randn(y.shape, generator=gen)
        y.backward(gy)
        cases.append(dict(x=_np(x), w=_np(w), b=_np(b), eps=1e-5, y=_np(y), gy=_np(gy), dx=_np(x.grad), dw=_np(w.grad), db=_np(b.grad)))
    m = SwishLayerNorm(512)
    m.set_hammer_kernel(PT)
    with torch.no_grad():
        m.weight.copy_(1 + 0.1 * torch.randn(512, generator=gen))
        m.bias.copy_(0.1 * torch.randn(512, generator=gen))
    x = torch.randn(9, 512, generator=gen)
    cases.append(dict(x=_np(x), w=_np(m.weight), b=_np(m.bias), eps=1e-5, y=_np(m(x)), module=1))
    return cases


def stu_case():
    torch.manual_seed(1234)
    gen = torch.Generator().manual_seed(4321)
    B, D, H, A, Hd, N = 4, 32, 2, 16, 24, 30
    layers = []
    for gn in (False, True):
        cfg = STULayerConfig(embedding_dim=D, num_heads=H, hidden_dim=Hd, attention_dim=A,
                             output_dropout_ratio=0.0, causal=True, target_aware=True, max_attn_len=None,
                             attn_alpha=None, use_group_norm=gn, recompute_normed_x=True,
                             recompute_uvqk=True, recompute_y=True, sort_by_length=True, contextual_seq_len=0)
        layers.append(STULayer(cfg, is_inference=False))
    stack = STUStack(layers, is_inference=False)
    stack.set_hammer_kernel(PT)
    for p in stack.parameters():  # make biases / norm params non-trivial
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn(p.shape, generator=gen))
    nt = torch.randint(1, 4, (B,), generator=gen)
    lengths = torch.randint(0, N - 4, (B,), generator=gen) + nt
    offsets = torch.zeros(B + 1, dtype=torch.int64); offsets[1:] = torch.cumsum(lengths, 0)
    x = torch.randn(int(offsets[-1]), D, generator=gen).requires_grad_()
    y = stack(x=x, x_lengths=lengths, x_offsets=offsets, max_seq_len=N, num_targets=nt)
    gy = torch.randn(y.shape, generator=gen)
    y.backward(gy)
    d = dict(D=D, H=H, A=A, Hd=Hd, N=N, lengths=_np(lengths), offsets=_np(offsets), num_targets=_np(nt),
             x=_np(x), y=_np(y), gy=_np(gy), dx=_np(x.grad))
    for name, p in stack.named_parameters():
        d["p:" + name] = _np(p)
        d["g:" + name] = _np(p.grad)
    return d


def research_case():
    """Research-path attention with relative position + bucketed time bias
    (research/modeling/sequential/hstu.py:87-223)."""
    from generative_recommenders.research.modeling.sequential.hstu import (
        RelativeBucketedTimeAndPositionBasedBias,
        _hstu_attention_maybe_from_cache,
    )
    gen = torch.Generator().manual_seed(31)
    B, H, A, Ld, n = 4, 2, 16, 16, 24
    lengths = torch.randint(1, n + 1, (B,), generator=gen)
    lengths[0] = n
    offsets = torch.zeros(B + 1, dtype=torch.int64); offsets[1:] = torch.cumsum(lengths, 0)
    Lt = int(offsets[-1])
    torch.manual_seed(3)
    bias = RelativeBucketedTimeAndPositionBasedBias(
        max_seq_len=n, num_buckets=128,
        bucketization_fn=lambda x: (torch.log(torch.abs(x).clamp(min=1)) / 0.301).long())
    ts = torch.sort(torch.randint(0, 10**8, (B, n), generator=gen), dim=1).values
    q = (0.3 * torch.randn(Lt, H * A, generator=gen)).requires_grad_()
    k = (0.3 * torch.randn(Lt, H * A, generator=gen)).requires_grad_()
    v = (0.3 * torch.randn(Lt, H * Ld, generator=gen)).requires_grad_()
    mask = 1.0 - torch.triu(torch.ones(n, n), diagonal=1)
    out, _, _ = _hstu_attention_maybe_from_cache(
        num_heads=H, attention_dim=A, linear_dim=Ld, q=q, k=k, v=v, cached_q=None, cached_k=None,
        delta_x_offsets=None, x_offsets=offsets, all_timestamps=ts, invalid_attn_mask=mask,
        rel_attn_bias=bias)
    g = torch.randn(out.shape, generator=gen)
    out.backward(g)
    return dict(n=n, H=H, A=A, Ld=Ld, offsets=_np(offsets), ts=_np(ts), q=_np(q), k=_np(k), v=_np(v),
                pos_w=_np(bias._pos_w), ts_w=_np(bias._ts_w), out=_np(out), g=_np(g), dq=_np(q.grad),
                dk=_np(k.grad), dv_=_np(v.grad), dpos_w=_np(bias._pos_w.grad), dts_w=_np(bias._ts_w.grad))


def position_cases():
    """add_timestamp_positional_embeddings, PYTORCH branch (ops/position.py:38-96, ops/pytorch/pt_position.py:40-134):
    the step right before the STU stack (modules/positional_encoder.py:52-75)."""
    from generative_recommenders.ops.position import add_timestamp_positional_embeddings
    cases = []
    for seed, (B, N, D, ctx, targets, interleave, fn, npos, ntime) in enumerate([
            (6, 40, 32, 0, True, False, "sqrt", 64, 48),
            (5, 33, 16, 3, True, True, "log", 30, 48),       # position table smaller than N: index clamp
            (4, 50, 64, 0, False, False, "sqrt", 128, 100)]):
        gen = torch.Generator().manual_seed(100 + seed)
        lengths = torch.randint(ctx + 4, N + 1, (B,), generator=gen)
        lengths[0] = N
        if not targets:   # the reference's PyTorch branch indexes the position table with L - col < 0 for padded
            lengths[:] = N  # columns when there are no targets (IndexError): only full-length batches run there
        nt = torch.minimum(torch.randint(1, 4, (B,), generator=gen), (lengths - ctx) // 2) if targets else None
        offsets = torch.zeros(B + 1, dtype=torch.int64); offsets[1:] = torch.cumsum(lengths, 0)
        Lt = int(offsets[-1])
        ts = torch.cat([torch.sort(torch.randint(0, 3 * 10**6, (int(l),), generator=gen)).values for l in lengths])
        x = torch.randn(Lt, D, generator=gen).requires_grad_()
        pos_w = (0.1 * torch.randn(npos, D, generator=gen)).requires_grad_()
        ts_w = (0.1 * torch.randn(ntime + 1, D, generator=gen)).requires_grad_()
        alpha = float(D) ** 0.5
        out = add_timestamp_positional_embeddings(
            alpha=alpha, max_seq_len=N, max_contextual_seq_len=ctx, position_embeddings_weight=pos_w,
            timestamp_embeddings_weight=ts_w, seq_offsets=offsets, seq_lengths=lengths, seq_embeddings=x,
            timestamps=ts, num_targets=nt, interleave_targets=interleave, time_bucket_fn=fn, kernel=PT)
        g = torch.randn(out.shape, generator=gen)
        out.backward(g)
        cases.append(dict(N=N, D=D, ctx=ctx, interleave=int(interleave), fn=np.asarray(fn), alpha=alpha,
                          offsets=_np(offsets), lengths=_np(lengths), num_targets=None if nt is None else _np(nt),
                          ts=_np(ts), x=_np(x), pos_w=_np(pos_w), ts_w=_np(ts_w), out=_np(out), g=_np(g),
                          dx=_np(x.grad), dpos_w=_np(pos_w.grad), dts_w=_np(ts_w.grad)))
    return cases


def postprocess_cases():
    """Output postprocessors (modules/postprocessors.py:55-103) and HSTUTransducer._postprocess
    (modules/hstu_transducer.py:191-251: candidate split + postprocessor), fp32."""
    import types
    from generative_recommenders.modules.hstu_transducer import HSTUTransducer
    from generative_recommenders.modules.postprocessors import L2NormPostprocessor, LayerNormPostprocessor
    gen = torch.Generator().manual_seed(77)
    B, N, D = 6, 24, 32
    lengths = torch.randint(3, N + 1, (B,), generator=gen)
    nt = torch.minimum(torch.randint(1, 5, (B,), generator=gen), lengths - 1)
    Lt = int(lengths.sum())
    x = torch.randn(Lt, D, generator=gen)
    x[1] = 0.0                       # a zero row: the clamp(min=1e-6) branch of the l2 norm
    x[2] = 1e-9 * torch.randn(D, generator=gen)
    ts = torch.randint(0, 10**6, (Lt,), generator=gen)
    g = torch.randn(Lt, D, generator=gen)
    out = {}
    xl = x.clone().requires_grad_()
    yl = L2NormPostprocessor()(xl, ts, {})
    yl.backward(g)
    out.update(l2_out=_np(yl), l2_dx=_np(xl.grad))
    torch.manual_seed(5)
    lnp = LayerNormPostprocessor(embedding_dim=D, eps=1e-5)
    with torch.no_grad():
        lnp._layer_norm.weight.uniform_(0.5, 1.5)
        lnp._layer_norm.bias.uniform_(-0.5, 0.5)
    xn = x.clone().requires_grad_()
    yn = lnp(xn, ts, {})
    yn.backward(g)
    out.update(ln_w=_np(lnp._layer_norm.weight), ln_b=_np(lnp._layer_norm.bias), ln_out=_np(yn), ln_dx=_np(xn.grad),
               ln_dw=_np(lnp._layer_norm.weight.grad), ln_db=_np(lnp._layer_norm.bias.grad))
    # _postprocess with a stand-in self carrying exactly the attributes the method reads
    for full in (False, True):
        fake = types.SimpleNamespace(
            _return_full_embeddings=full, _output_postprocessor=L2NormPostprocessor(),
            _input_preprocessor=types.SimpleNamespace(interleave_targets=lambda: False),
            hammer_kernel=lambda: PT)
        xs = x.clone().requires_grad_()
        full_emb, cand = HSTUTransducer._postprocess(
            fake, max_seq_len=N, total_uih_len=int((lengths - nt).sum()), total_targets=int(nt.sum()), seq_lengths=lengths,
            seq_timestamps=ts, seq_embeddings=xs, num_targets=nt, seq_payloads={})
        gc = torch.randn(cand.shape, generator=torch.Generator().manual_seed(9))
        cand.backward(gc)
        tag = "full" if full else "cand"
        out.update({f"pp_{tag}_cand": _np(cand), f"pp_{tag}_gc": _np(gc), f"pp_{tag}_dx": _np(xs.grad)})
        if full:
            out["pp_full_emb"] = _np(full_emb)
    out.update(N=N, D=D, lengths=_np(lengths), num_targets=_np(nt), x=_np(x), ts=_np(ts), g=_np(g))
    return [out]


def sampled_softmax_cases():
    """SampledSoftmaxLoss (research/modeling/sequential/losses/sampled_softmax.py:44-193) with the dot-product
    similarity (rails/similarities/dot_product_similarity_fn.py) and both negatives samplers
    (autoregressive_losses.py:73-204), fp32.  The sampler's draw is recovered by re-seeding: its randint is the first
    RNG use inside jagged_forward."""
    import types
    from generative_recommenders.research.modeling.sequential.autoregressive_losses import (
        InBatchNegativesSampler, LocalNegativesSampler)
    from generative_recommenders.research.modeling.sequential.losses.sampled_softmax import SampledSoftmaxLoss
    from generative_recommenders.research.rails.similarities.dot_product_similarity_fn import DotProductSimilarity

    dp = DotProductSimilarity()
    model = types.SimpleNamespace(similarity_fn=lambda query_embeddings, item_ids, item_embeddings=None, **kw: dp(
        query_embeddings=query_embeddings, item_embeddings=item_embeddings, item_ids=item_ids, **kw))
    cases = []
    # ---- local sampler (the Amazon-Books configuration: l2 norm, T = 0.05) and without norm / temperature
    for ci, (l2, T, D, V, R, n_rows) in enumerate([(True, 0.05, 16, 40, 12, 20), (False, 1.0, 24, 15, 7, 11),
                                                   (True, 0.05, 64, 300, 512, 9)]):
        gen = torch.Generator().manual_seed(100 + ci)
        all_ids = (torch.randperm(5 * V, generator=gen)[:V] + 1).tolist()       # sparse id space, 0 = padding
        emb = torch.nn.Embedding(5 * V + 1, D)
        with torch.no_grad():
            emb.weight.copy_(torch.randn(5 * V + 1, D, generator=gen) * 0.5)
            emb.weight[all_ids[0]] = 0.0                                        # a zero row: the clamp branch
        sampler = LocalNegativesSampler(num_items=V, item_emb=emb, all_item_ids=all_ids, l2_norm=l2, l2_norm_eps=1e-6)
        loss_mod = SampledSoftmaxLoss(num_to_sample=R, softmax_temperature=T, model=model)
        pos_ids = torch.tensor(all_ids)[torch.randint(0, V, (n_rows,), generator=gen)]
        q = (torch.randn(n_rows, D, generator=gen) * 0.7).requires_grad_()
        pos_emb = emb(pos_ids).detach().clone().requires_grad_()
        w = (torch.rand(n_rows, generator=gen) > 0.25).float()
        seed = 4242 + ci
        torch.manual_seed(seed)
        loss, _ = loss_mod.jagged_forward(output_embeddings=q, supervision_ids=pos_ids, supervision_embeddings=pos_emb,
                                          supervision_weights=w, negatives_sampler=sampler)
        loss.backward()
        torch.manual_seed(seed)
        sampled_ids, _ = sampler(positive_ids=pos_ids, num_to_sample=R)
        cases.append(dict(kind=np.asarray("local"), l2=int(l2), T=T, R=R, eps=1e-6, all_item_ids=np.asarray(all_ids),
                          table=_np(emb.weight), q=_np(q), pos_emb=_np(pos_emb), pos_ids=_np(pos_ids), weights=_np(w),
                          sampled_ids=_np(sampled_ids), n_collisions=int((sampled_ids == pos_ids[:, None]).sum()),
                          loss=_np(loss), dq=_np(q.grad), dpos_emb=_np(pos_emb.grad), dtable=_np(emb.weight.grad)))
    # ---- in-batch sampler with de-duplication
    gen = torch.Generator().manual_seed(321)
    B, N, D, R = 5, 9, 16, 10
    ids = torch.randint(1, 14, (B, N), generator=gen)
    presences = torch.rand(B, N, generator=gen) > 0.2
    embs = (torch.randn(B, N, D, generator=gen) * 0.5).requires_grad_()
    sampler = InBatchNegativesSampler(l2_norm=True, l2_norm_eps=1e-6, dedup_embeddings=True)
    sampler.process_batch(ids=ids, presences=presences, embeddings=embs)
    n_rows = 13
    pos_ids = ids[presences][:n_rows].clone()
    q = (torch.randn(n_rows, D, generator=gen) * 0.7).requires_grad_()
    pos_emb = (torch.randn(n_rows, D, generator=gen) * 0.5).requires_grad_()
    w = torch.ones(n_rows)
    loss_mod = SampledSoftmaxLoss(num_to_sample=R, softmax_temperature=0.05, model=model)
    torch.manual_seed(99)
    loss, _ = loss_mod.jagged_forward(output_embeddings=q, supervision_ids=pos_ids, supervision_embeddings=pos_emb,
                                      supervision_weights=w, negatives_sampler=sampler)
    loss.backward()
    torch.manual_seed(99)
    X = sampler._cached_ids.size(0)
    offsets = torch.randint(low=0, high=X, size=(n_rows, R), dtype=pos_ids.dtype)
    cases.append(dict(kind=np.asarray("in-batch"), l2=1, T=0.05, R=R, eps=1e-6, ids=_np(ids), presences=_np(presences),
                      embeddings=_np(embs), cached_ids=_np(sampler._cached_ids), cached_embeddings=_np(sampler._cached_embeddings),
                      sampled_offsets=_np(offsets), q=_np(q), pos_emb=_np(pos_emb), pos_ids=_np(pos_ids), weights=_np(w),
                      loss=_np(loss), dq=_np(q.grad), dpos_emb=_np(pos_emb.grad), dembeddings=_np(embs.grad)))
    # ---- the padded entry point forward(lengths, (B, N, D) ...) -> dense_to_jagged -> jagged_forward
    gen = torch.Generator().manual_seed(555)
    B, N, D, V, R = 4, 7, 16, 30, 6
    all_ids = list(range(1, V + 1))
    emb = torch.nn.Embedding(V + 1, D)
    with torch.no_grad():
        emb.weight.copy_(torch.randn(V + 1, D, generator=gen) * 0.5)
    sampler = LocalNegativesSampler(num_items=V, item_emb=emb, all_item_ids=all_ids, l2_norm=True, l2_norm_eps=1e-6)
    lengths = torch.tensor([7, 0, 3, 5])
    sup_ids = torch.randint(1, V + 1, (B, N), generator=gen)
    out_emb = (torch.randn(B, N, D, generator=gen) * 0.7).requires_grad_()
    sup_emb = emb(sup_ids).detach().clone().requires_grad_()
    sw = torch.rand(B, N, generator=gen)
    loss_mod = SampledSoftmaxLoss(num_to_sample=R, softmax_temperature=0.05, model=model)
    torch.manual_seed(7)
    loss, _ = loss_mod(lengths=lengths, output_embeddings=out_emb, supervision_ids=sup_ids, supervision_embeddings=sup_emb,
                       supervision_weights=sw, negatives_sampler=sampler)
    loss.backward()
    off = torch.ops.fbgemm.asynchronous_complete_cumsum(lengths)
    jag_ids = torch.ops.fbgemm.dense_to_jagged(sup_ids.unsqueeze(-1).float(), [off])[0].squeeze(1).long()
    torch.manual_seed(7)
    sampled_ids, _ = sampler(positive_ids=jag_ids, num_to_sample=R)
    cases.append(dict(kind=np.asarray("local-padded"), l2=1, T=0.05, R=R, eps=1e-6, all_item_ids=np.asarray(all_ids),
                      table=_np(emb.weight), lengths=_np(lengths), sup_ids=_np(sup_ids), out_emb=_np(out_emb),
                      sup_emb=_np(sup_emb), sup_weights=_np(sw), sampled_ids=_np(sampled_ids), loss=_np(loss),
                      dout_emb=_np(out_emb.grad), dsup_emb=_np(sup_emb.grad), dtable=_np(emb.weight.grad)))
    return cases


def _bf16_bits(t):
    """bf16-representable fp32 tensor -> its 16-bit patterns (half the bytes in the fixture)"""
    return t.detach().to(torch.bfloat16).view(torch.int16).numpy()


def metric_shape_cases():
    """hstu_mha forward + backward at the shapes the throughput is quoted on: M = N 200, 4 heads of 128
    (the folded backward kernel's shape) and C2 = N 211, 4 heads of 64, with targets.  The inputs are drawn in fp32 and
    ROUNDED TO BF16 before the reference (fp32 arithmetic) sees them, so one fixture serves three checks: the fp32
    kernels on the same values, the bf16 kernels against the exact answer for their own inputs, and -- against the
    bf16-rounded reference outputs -- the kernels' error without the unavoidable output rounding.  Inputs are stored as
    bf16 bit patterns."""
    cases = []
    for ci, (name, lens, N, H, d, targets) in enumerate([("M", [200, 77], 200, 4, 128, None),
                                                          ("C2", [211, 100], 211, 4, 64, [6, 3])]):
        gen = torch.Generator().manual_seed(9000 + ci)
        lengths = torch.tensor(lens)
        offsets = torch.zeros(len(lens) + 1, dtype=torch.int64); offsets[1:] = torch.cumsum(lengths, 0)
        Lt = int(offsets[-1])
        r = lambda t: t.to(torch.bfloat16).to(torch.float32)
        q = r(torch.randn(Lt, H, d, generator=gen)).requires_grad_()
        k = r(torch.randn(Lt, H, d, generator=gen)).requires_grad_()
        v = r(0.5 * torch.randn(Lt, H, d, generator=gen)).requires_grad_()
        dout = r(0.1 * torch.randn(Lt, H, d, generator=gen))
        nt = None if targets is None else torch.tensor(targets)
        alpha = 1.0 / (d**0.5)
        out = hstu_mha(max_seq_len=N, alpha=alpha, q=q, k=k, v=v, seq_offsets=offsets, causal=True, dropout_pr=0.0,
                       training=False, num_targets=nt, max_attn_len=0, contextual_seq_len=0, min_full_attn_seq_len=0,
                       kernel=PT)
        out.backward(dout)
        cases.append(dict(name=np.asarray(name), N=N, H=H, d=d, alpha=alpha, offsets=_np(offsets),
                          num_targets=None if nt is None else _np(nt), q_bf16=_bf16_bits(q), k_bf16=_bf16_bits(k),
                          v_bf16=_bf16_bits(v), dout_bf16=_bf16_bits(dout), out=_np(out), dq=_np(q.grad), dk=_np(k.grad),
                          dv_=_np(v.grad)))
    return cases


def research_layer_cases():
    """The research-path layer end to end: two ``SequentialTransductionUnitJagged`` layers inside ``HSTUJagged``
    (research/modeling/sequential/hstu.py:226-540), forward + backward with every parameter gradient, the cache states
    (``return_cache_states``), the INCREMENTAL call (``delta_x_offsets`` / ``cache``: one new last row per user) and the
    ``all_timestamps=None`` call (no relative bias at all, :205-206)."""
    from generative_recommenders.research.modeling.sequential.hstu import (
        HSTUJagged,
        RelativeBucketedTimeAndPositionBasedBias,
        SequentialTransductionUnitJagged,
    )
    cases = []
    for ci, concat_ua in enumerate([False, True]):
        gen = torch.Generator().manual_seed(700 + ci)
        B, n, D, H, A, Ld = 3, 24, 32, 2, 16, 16
        torch.manual_seed(11 + ci)
        layers = [SequentialTransductionUnitJagged(
            embedding_dim=D, linear_hidden_dim=Ld, attention_dim=A, dropout_ratio=0.0, attn_dropout_ratio=0.0, num_heads=H,
            linear_activation="silu", relative_attention_bias_module=RelativeBucketedTimeAndPositionBasedBias(
                max_seq_len=n, num_buckets=128,
                bucketization_fn=lambda x: (torch.log(torch.abs(x).clamp(min=1)) / 0.301).long()),
            normalization="rel_bias", linear_config="uvqk", concat_ua=concat_ua, epsilon=1e-6) for _ in range(2)]
        model = HSTUJagged(layers, autocast_dtype=None)
        with torch.no_grad():
            for prm in model.parameters():   # larger than the 0.02 init so that every term matters
                prm.add_(0.05 * torch.randn(prm.shape, generator=gen))
        lengths = torch.randint(2, n + 1, (B,), generator=gen)
        lengths[0] = n
        offsets = torch.zeros(B + 1, dtype=torch.int64); offsets[1:] = torch.cumsum(lengths, 0)
        Lt = int(offsets[-1])
        ts = torch.sort(torch.randint(0, 10**8, (B, n), generator=gen), dim=1).values
        mask = 1.0 - torch.triu(torch.ones(n, n), diagonal=1)
        x = torch.randn(Lt, D, generator=gen).requires_grad_()
        y, cache = model.jagged_forward(x=x, x_offsets=offsets, all_timestamps=ts, invalid_attn_mask=mask,
                                        return_cache_states=True)
        g = torch.randn(y.shape, generator=gen)
        y.backward(g)
        d = dict(concat_ua=int(concat_ua), B=B, n=n, D=D, H=H, A=A, Ld=Ld, offsets=_np(offsets), ts=_np(ts), x=_np(x),
                 y=_np(y), g=_np(g), dx=_np(x.grad))
        for name, prm in model.named_parameters():
            d["p:" + name] = _np(prm)
            d["g:" + name] = _np(prm.grad)
        for li, (cv, cq, ck, co) in enumerate(cache):
            d.update({f"cache{li}:v": _np(cv), f"cache{li}:q": _np(cq), f"cache{li}:k": _np(ck), f"cache{li}:out": _np(co)})
        # incremental: every user's LAST row is replaced by a new item; the caches of the call above stand for the
        # state before it (their last rows are overwritten by the call, as the reference's index_copy_ does)
        with torch.no_grad():
            rows = offsets[1:] - 1
            cols = lengths - 1
            x2 = x.detach().clone()
            x2[rows] = torch.randn(B, D, generator=gen)
            cache_in = [tuple(t.detach().clone() for t in c) for c in cache]
            y2, cache2 = model.jagged_forward(x=x2, x_offsets=offsets, all_timestamps=ts, invalid_attn_mask=mask,
                                              delta_x_offsets=(rows, cols), cache=cache_in, return_cache_states=True)
            y_full, _ = model.jagged_forward(x=x2, x_offsets=offsets, all_timestamps=ts, invalid_attn_mask=mask)
            y_nobias, _ = model.jagged_forward(x=x.detach(), x_offsets=offsets, all_timestamps=None, invalid_attn_mask=mask)
        d.update(x2=_np(x2), delta_rows=_np(rows), delta_cols=_np(cols), y2=_np(y2), y2_full=_np(y_full), y_nobias=_np(y_nobias))
        for li, (cv, cq, ck, co) in enumerate(cache2):
            d.update({f"cache2_{li}:v": _np(cv), f"cache2_{li}:q": _np(cq), f"cache2_{li}:k": _np(ck), f"cache2_{li}:out": _np(co)})
        cases.append(d)
    return cases


class _StubEmb(torch.nn.Module):
    """stand-in EmbeddingModule of the top-level HSTU model (the test rebuilds it with the same values)"""
    item_embedding_dim = 32

    def __init__(self, n_items=50, dim=32):
        super().__init__()
        self._item_emb = torch.nn.Embedding(n_items, dim)

    def get_item_embeddings(self, ids):
        return self._item_emb(ids)


class _StubPre(torch.nn.Module):
    """stand-in InputFeaturesPreprocessorModule: scaled item embeddings + a learned positional table, padded rows zeroed"""

    def __init__(self, n, dim):
        super().__init__()
        self._pos = torch.nn.Parameter(0.1 * torch.randn(n, dim))

    def forward(self, past_lengths, past_ids, past_embeddings, past_payloads):
        B, N, D = past_embeddings.shape
        x = past_embeddings * (D ** 0.5) + self._pos[:N].unsqueeze(0)
        valid = (past_ids != 0).unsqueeze(-1).to(x.dtype)
        return past_lengths, x * valid, valid


class _StubPost(torch.nn.Module):
    """stand-in OutputPostprocessorModule: L2 normalisation (research/modeling/sequential/output_postprocessors.py)"""

    def forward(self, x):
        return x / torch.clamp(torch.linalg.norm(x, ord=None, dim=-1, keepdim=True), min=1e-6)


class _StubSim(torch.nn.Module):
    def forward(self, query_embeddings, item_embeddings, item_ids=None, **kw):
        return (query_embeddings.unsqueeze(1) * item_embeddings).sum(-1), {}


def hstu_model_cases():
    """The top-level research model ``HSTU`` (research/modeling/sequential/hstu.py:543-809) with stand-in embedding /
    preprocessor / postprocessor / similarity modules (defined above; tests/test_research_gpu.py rebuilds them with the
    saved values): ``forward`` (B, N, D), ``encode`` (B, D), every parameter gradient of a loss on both."""
    from generative_recommenders.research.modeling.sequential.hstu import HSTU

    cases = []
    for ci, concat_ua in enumerate([False, True]):
        gen = torch.Generator().manual_seed(900 + ci)
        B, N, D, H, A, Ld, out_len = 3, 20, 32, 2, 16, 16, 4
        torch.manual_seed(21 + ci)
        model = HSTU(max_sequence_len=N, max_output_len=out_len, embedding_dim=D, num_blocks=2, num_heads=H, linear_dim=Ld,
                     attention_dim=A, normalization="rel_bias", linear_config="uvqk", linear_activation="silu",
                     linear_dropout_rate=0.0, attn_dropout_rate=0.0, embedding_module=_StubEmb(50, D),
                     similarity_module=_StubSim(), input_features_preproc_module=_StubPre(N + out_len, D),
                     output_postproc_module=_StubPost(), concat_ua=concat_ua, verbose=False)
        with torch.no_grad():
            for prm in model.parameters():
                prm.add_(0.05 * torch.randn(prm.shape, generator=gen))
        Nt = N + out_len
        lengths = torch.tensor([Nt, 7, 13])
        ids = torch.randint(1, 50, (B, Nt), generator=gen)
        ids = ids * (torch.arange(Nt).unsqueeze(0) < lengths.unsqueeze(1))     # padded positions: id 0
        ts = torch.sort(torch.randint(0, 10**7, (B, Nt), generator=gen), dim=1).values
        emb = model.get_item_embeddings(ids)
        y = model(past_lengths=lengths, past_ids=ids, past_embeddings=emb, past_payloads={"timestamps": ts})
        cur = model.encode(past_lengths=lengths, past_ids=ids, past_embeddings=emb, past_payloads={"timestamps": ts})
        gy = torch.randn(y.shape, generator=gen)
        gc = torch.randn(cur.shape, generator=gen)
        ((y * gy).sum() + (cur * gc).sum()).backward()
        d = dict(concat_ua=int(concat_ua), B=B, N=N, out_len=out_len, D=D, H=H, A=A, Ld=Ld, lengths=_np(lengths), ids=_np(ids),
                 ts=_np(ts), y=_np(y), cur=_np(cur), gy=_np(gy), gc=_np(gc))
        for name, prm in model.named_parameters():
            d["p:" + name] = _np(prm)
            d["g:" + name] = _np(prm.grad)
        cases.append(d)
    return cases


def _save_cases(path, cases):
    flat = {}
    for i, c in enumerate(cases):
        for key, val in c.items():
            if val is None:
                continue
            flat[f"c{i}:{key}"] = np.asarray(val)
    flat["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(path, **flat)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE, help="directory the .npz files are written to (default: next to this script)")
    ap.add_argument("--only", default="", help="comma-separated fixture names (without .npz); default: all")
    args = ap.parse_args()
    torch.set_num_threads(1)
    OUT = args.out
    os.makedirs(OUT, exist_ok=True)
    only = set(filter(None, args.only.split(",")))
    jobs = [
        ("attention", attention_cases), ("delta_attention", delta_cases), ("jagged", lambda: jagged_cases()[0]),
        ("jagged_l2", lambda: [jagged_cases()[1]]),
        ("compute", lambda: [dict(name=np.asarray(n), **c) for n, c in compute_cases().items()]),
        ("swish_layer_norm", swish_layer_norm_cases),
        ("stu", lambda: [stu_case()]), ("research_attention", lambda: [research_case()]), ("position", position_cases),
        ("postprocess", postprocess_cases), ("sampled_softmax", sampled_softmax_cases),
        ("metric_shapes", metric_shape_cases), ("research_layer", research_layer_cases), ("hstu_model", hstu_model_cases),
    ]
    for name, fn in jobs:
        if only and name not in only:
            continue
        _save_cases(os.path.join(OUT, name + ".npz"), fn())
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
