"""GPU parity of the output post-processing: row L2 norm / layer-norm postprocessors and the
candidate split of HSTUTransducer._postprocess, against the reference's golden vectors and the oracle.
fp32 I/O: 1e-5 relative (the gradient through a ~1e-9 row divides by its norm: absolute slack 1e-4)."""

import numpy as np
import pytest
import torch

from conftest import load_cases
from oracle import hstu_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.requires_grad_() if grad else t


def test_golden_postprocessors():
    from generative_recommenders_amd.modules.postprocessors import L2NormPostprocessor, LayerNormPostprocessor

    c = load_cases("postprocess.npz")[0]
    x = _t(c["x"], True)
    y = L2NormPostprocessor()(x, _t(c["ts"]), {})
    np.testing.assert_allclose(y.detach().cpu().numpy(), c["l2_out"], rtol=1e-5, atol=1e-6)
    y.backward(_t(c["g"]))
    np.testing.assert_allclose(x.grad.cpu().numpy(), c["l2_dx"], rtol=2e-5, atol=1e-4)
    lnp = LayerNormPostprocessor(embedding_dim=int(c["D"]), eps=1e-5).to(DEV)
    assert sorted(lnp.state_dict()) == ["_layer_norm.bias", "_layer_norm.weight"]
    with torch.no_grad():
        lnp._layer_norm.weight.copy_(_t(c["ln_w"]))
        lnp._layer_norm.bias.copy_(_t(c["ln_b"]))
    xn = _t(c["x"], True)
    yn = lnp(xn, _t(c["ts"]), {})
    np.testing.assert_allclose(yn.detach().cpu().numpy(), c["ln_out"], rtol=1e-4, atol=1e-5)
    yn.backward(_t(c["g"]))
    np.testing.assert_allclose(xn.grad.cpu().numpy(), c["ln_dx"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(lnp._layer_norm.weight.grad.cpu().numpy(), c["ln_dw"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lnp._layer_norm.bias.grad.cpu().numpy(), c["ln_db"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("full", [False, True])
def test_golden_candidate_split_and_postprocess(full):
    from generative_recommenders_amd.modules.hstu_transducer import hstu_postprocess
    from generative_recommenders_amd.modules.postprocessors import L2NormPostprocessor

    c = load_cases("postprocess.npz")[0]
    lengths, nt = c["lengths"], c["num_targets"]
    x = _t(c["x"], True)
    emb, cand = hstu_postprocess(L2NormPostprocessor(), max_seq_len=int(c["N"]), total_uih_len=int((lengths - nt).sum()),
                                 total_targets=int(nt.sum()), seq_lengths=_t(lengths), seq_timestamps=_t(c["ts"]),
                                 seq_embeddings=x, num_targets=_t(nt), seq_payloads={}, return_full_embeddings=full)
    tag = "full" if full else "cand"
    np.testing.assert_allclose(cand.detach().cpu().numpy(), c[f"pp_{tag}_cand"], rtol=1e-5, atol=1e-6)
    assert (emb is None) == (not full)
    if full:
        np.testing.assert_allclose(emb.detach().cpu().numpy(), c["pp_full_emb"], rtol=1e-5, atol=1e-6)
    cand.backward(_t(c[f"pp_{tag}_gc"]))
    np.testing.assert_allclose(x.grad.cpu().numpy(), c[f"pp_{tag}_dx"], rtol=2e-5, atol=1e-4)


@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 512), (torch.float32, 100), (torch.float16, 64), (torch.bfloat16, 1024)])
def test_l2_norm_vs_oracle(dtype, D):
    from generative_recommenders_amd.modules.postprocessors import l2_norm

    rng = np.random.default_rng(D)
    x = torch.from_numpy(rng.standard_normal((777, D))).to(dtype)
    x[5] = 0
    g = torch.from_numpy(rng.standard_normal((777, D))).to(dtype)
    xd = x.to(DEV).requires_grad_()
    y = l2_norm(xd)
    y.backward(g.to(DEV))
    ref = O.l2_norm_fwd(x.double().numpy())
    rdx = O.l2_norm_bwd(g.double().numpy(), x.double().numpy())
    if dtype == torch.float32:
        np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(xd.grad.cpu().numpy(), rdx, rtol=1e-4, atol=1e-5)
    else:
        np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref, rtol=1.6e-2, atol=1e-3)
        ok = np.ones(777, dtype=bool)
        ok[5] = False          # g / eps = 1e6 * g overflows nothing but is far outside any 16-bit tolerance band
        np.testing.assert_allclose(xd.grad.float().cpu().numpy()[ok], rdx[ok], rtol=3e-2, atol=3e-3)
