"""CPU tests of the drop-in boundary: the C-ABI library builds/loads, exports every symbol
include/hstu_hip.h declares, the ctypes mirrors of the parameter structs have the C layout,
and the host-side API raises the reference's errors.  No kernel is launched."""

import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hstu_hip.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hstu_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from generative_recommenders_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_symbols_all_exported(lib):
    declared = _declared_symbols()
    assert len(declared) >= 26
    for name in declared:
        assert hasattr(lib, name), f"libhstu_hip.so does not export {name}"


def test_ctypes_signature_table_covers_header():
    from generative_recommenders_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_abi_version_and_error_string(lib):
    from generative_recommenders_amd import _lib as L
    assert lib.hstu_abi_version() == L.ABI_VERSION == 13
    assert isinstance(lib.hstu_last_error(), bytes)


def test_struct_layout_matches_c():
    from generative_recommenders_amd import _lib

    prog = r"""
#include <stdio.h>
#include <stddef.h>
#include "hstu_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(HstuAttnParams), offsetof(HstuAttnParams, batch),
         offsetof(HstuAttnParams, alpha), offsetof(HstuAttnParams, dtype), sizeof(HstuAttnBwdParams),
         offsetof(HstuAttnBwdParams, dout), offsetof(HstuAttnBwdParams, workspace),
         offsetof(HstuAttnBwdParams, total_rows), offsetof(HstuAttnParams, pos_w),
         offsetof(HstuAttnParams, bucket_div), offsetof(HstuAttnBwdParams, dts_w));
  return 0;
}
"""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "layout.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "layout")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    P, BP = _lib.HstuAttnParams, _lib.HstuAttnBwdParams
    exp = [C.sizeof(P), P.batch.offset, P.alpha.offset, P.dtype.offset, C.sizeof(BP), BP.dout.offset,
           BP.workspace.offset, BP.total_rows.offset, P.pos_w.offset, P.bucket_div.offset, BP.dts_w.offset]
    assert got == exp


def test_validation_errors_without_gpu(lib):
    """Argument validation happens before any launch, so it is testable on CPU."""
    from generative_recommenders_amd import _lib

    p = _lib.HstuAttnParams()
    assert lib.hstu_attn_fwd(C.byref(p), None) == -1
    assert b"non-NULL" in lib.hstu_last_error()
    buf = (C.c_char * 256)()
    addr = (C.addressof(buf) + 15) & ~15
    p.q = p.k = p.v = p.out = p.seq_offsets = addr
    p.batch, p.heads, p.max_seq_len, p.dqk, p.dv = 1, 1, 0, 32, 32
    assert lib.hstu_attn_fwd(C.byref(p), None) == -1
    assert b"max_seq_len must be larger than 0" in lib.hstu_last_error()
    p.max_seq_len = 8
    p.dqk = 12
    assert lib.hstu_attn_fwd(C.byref(p), None) == -1 and b"multiples of 8" in lib.hstu_last_error()
    p.dqk = p.dv = 256
    p.q_row_stride = p.k_row_stride = p.v_row_stride = p.o_row_stride = 256
    p.q_head_stride = p.k_head_stride = p.v_head_stride = p.o_head_stride = 256
    assert lib.hstu_attn_fwd(C.byref(p), None) == -2 and b"not instantiated" in lib.hstu_last_error()
    assert lib.hstu_split_2d_jagged(None, None, None, None, None, 0, 0, 0, 1, 1, 4, 0, 0, None) == -1


def test_ln_linear_shape_rule_and_validation_without_gpu(lib):
    """hstu_ln_linear_fwd (ABI v9): which shapes the fused LayerNorm + projection kernel takes, and its argument checks --
    all before any launch"""
    BF16, F16, F32 = 0, 1, 2
    sup = lib.hstu_ln_linear_fwd_supported
    assert sup(1000, 512, 2048, BF16) == 1 and sup(1, 512, 32, F16) == 1 and sup(10**6, 512, 4096, BF16) == 1
    assert sup(1000, 512, 2048, F32) == 0            # fp32 activations: layer norm + GEMM
    assert sup(1000, 256, 2048, BF16) == 0 and sup(1000, 1024, 2048, BF16) == 0     # embedding dims other than 512
    assert sup(1000, 512, 2000, BF16) == 0 and sup(1000, 512, 4128, BF16) == 0 and sup(1000, 512, 0, BF16) == 0
    buf = (C.c_char * 4096)()
    a = (C.addressof(buf) + 15) & ~15
    call = lambda x=a, ldx=512, w=a, y=a, ldy=2048, normed=None, ldn=0, rows=4, k=512, n=2048, dt=BF16: lib.hstu_ln_linear_fwd(
        x, ldx, a, a, 1e-6, w, a, y, ldy, normed, ldn, None, None, rows, k, n, dt, None)
    assert call(rows=0) == 0                          # nothing to do: no launch, no error
    assert call(x=None) == -1 and b"NULL" in lib.hstu_last_error()
    assert call(k=256) == -1 and b"k == 512" in lib.hstu_last_error()
    assert call(dt=F32) == -1 and b"bf16 / fp16" in lib.hstu_last_error()
    assert call(ldy=2040) == -1 and b"leading dimension" in lib.hstu_last_error()
    assert call(ldx=516) == -1 and b"multiples of 8" in lib.hstu_last_error()
    assert call(y=a + 8) == -1 and b"16-byte aligned" in lib.hstu_last_error()
    assert call(normed=a, ldn=100) == -1 and b"leading dimension" in lib.hstu_last_error()


def test_linear_k512_shape_rule_and_validation_without_gpu(lib):
    """hstu_linear_k512 (ABI v11): the fused kernel's shape rule, and its argument checks -- all before any launch"""
    BF16, F32 = 0, 2
    sup = lib.hstu_linear_k512_supported
    assert sup(1000, 512, 1536, BF16) == 1 and sup(1000, 512, 1536, F32) == 0 and sup(1000, 256, 1536, BF16) == 0 and sup(5, 512, 1000, BF16) == 0
    buf = (C.c_char * 4096)()
    a = (C.addressof(buf) + 15) & ~15
    call = lambda x=a, ldx=512, w=a, y=a, ldy=1536, rows=4, k=512, n=1536, dt=BF16: lib.hstu_linear_k512(x, ldx, w, None, y, ldy, rows, k, n, dt, None)
    assert call(rows=0) == 0
    assert call(w=None) == -1 and b"NULL" in lib.hstu_last_error()
    assert call(k=1536, n=512) == -1 and b"k == 512" in lib.hstu_last_error()
    assert call(ldy=1528) == -1 and b"leading dimension" in lib.hstu_last_error()
    assert call(ldx=516) == -1 and b"multiples of 8" in lib.hstu_last_error()
    assert call(x=a + 8) == -1 and b"16-byte aligned" in lib.hstu_last_error()


def test_ops_fail_loudly_on_cpu_tensors():
    from generative_recommenders_amd.ops.hstu_attention import delta_hstu_mha, hstu_mha
    from generative_recommenders_amd.ops.jagged_tensors import asynchronous_complete_cumsum, concat_2D_jagged
    from generative_recommenders_amd.ops.layer_norm import layer_norm

    q = torch.zeros(4, 2, 32, dtype=torch.bfloat16)
    off = torch.tensor([0, 4])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hstu_mha(4, 1.0, q, q, q, off)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        delta_hstu_mha(4, 1.0, q, q, q, off)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        asynchronous_complete_cumsum(torch.tensor([1, 2]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        concat_2D_jagged(8, torch.zeros(4, 3), torch.zeros(4, 3), 4, 4, off, off)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer_norm(torch.zeros(4, 8), torch.ones(8), torch.zeros(8), 1e-6)


def test_reference_assert_messages():
    from generative_recommenders_amd.ops.hstu_attention import hstu_mha
    from generative_recommenders_amd.ops.jagged_tensors import split_2D_jagged

    q = torch.zeros(4, 2, 32)
    off = torch.tensor([0, 4])
    with pytest.raises(Exception, match="max_seq_len must be larger than 0"):
        hstu_mha(0, 1.0, q, q, q, off)
    with pytest.raises(Exception, match="k must be the same shape as q"):
        hstu_mha(4, 1.0, q, q[:, :1], q, off)
    with pytest.raises(Exception, match="only support causal"):
        hstu_mha(4, 1.0, q, q, q, off, causal=False)
    with pytest.raises(Exception, match="cannot be None at the same time"):
        split_2D_jagged(8, torch.zeros(4, 3))


def test_missing_library_raises(monkeypatch):
    from generative_recommenders_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libhstu_hip.so")
    with pytest.raises(_lib.HstuLibraryError, match="no CPU / PyTorch fallback"):
        _lib.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "generative_recommenders_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text, f"{f} mentions the oracle"


def test_module_state_dict_names_match_reference():
    from generative_recommenders_amd.modules.stu import STULayer, STULayerConfig

    layer = STULayer(STULayerConfig(embedding_dim=16, num_heads=2, hidden_dim=8, attention_dim=8, use_group_norm=True))
    assert sorted(layer.state_dict()) == sorted([
        "_uvqk_weight", "_uvqk_beta", "_input_norm_weight", "_input_norm_bias", "_output_weight",
        "_output_norm_weight", "_output_norm_bias"])
    assert layer._uvqk_weight.shape == (16, 64) and layer._output_weight.shape == (48, 16)
    assert layer._output_norm_weight.shape == (2,)


def test_sort_kv_pairs_operator_semantics():
    """hstu::sort_kv_pairs (ops/cpp/cpp_ops.cpp:101): stable, optional low-bit key range, both directions -- checked
    against a python sort on CPU tensors (the op is index plumbing on torch.sort and runs on any device)."""
    import torch
    from generative_recommenders_amd.ops import torch_library

    torch_library.register()
    g = torch.Generator().manual_seed(0)
    keys = torch.randint(-50, 50, (200,), generator=g)
    vals = torch.arange(200)
    for desc in (False, True):
        k, v = torch.ops.hstu.sort_kv_pairs(keys, vals, None, desc)
        want = sorted(range(200), key=lambda i: (-keys[i].item() if desc else keys[i].item(), i))
        assert v.tolist() == want and k.tolist() == [keys[i].item() for i in want]
        k4, v4 = torch.ops.hstu.sort_kv_pairs(keys, vals, 4, desc)
        sub = lambda i: keys[i].item() & 15
        want4 = sorted(range(200), key=lambda i: (-sub(i) if desc else sub(i), i))
        assert v4.tolist() == want4 and k4.tolist() == [keys[i].item() for i in want4]
    k0, v0 = torch.ops.hstu.sort_kv_pairs(keys.int(), vals.float(), 0, False)
    assert k0.tolist() == keys.tolist() and v0.dtype == torch.float32


def test_negatives_samplers_contract_and_no_cpu_path():
    """The samplers are index plumbing and run anywhere; the fused loss refuses CPU tensors (no fallback)."""
    import pytest
    import torch
    import generative_recommenders_amd.research.modeling.sequential.autoregressive_losses as AL
    import generative_recommenders_amd.research.modeling.sequential.losses.sampled_softmax as SS

    torch.manual_seed(0)
    emb = torch.nn.Embedding(20, 8)
    s = AL.LocalNegativesSampler(num_items=10, item_emb=emb, all_item_ids=list(range(1, 11)), l2_norm=True, l2_norm_eps=1e-6)
    ids, e = s(positive_ids=torch.tensor([1, 2, 3]), num_to_sample=4)
    assert ids.shape == (3, 4) and e.shape == (3, 4, 8) and ids.min() >= 1 and ids.max() <= 10
    torch.testing.assert_close(e.norm(dim=-1), torch.ones(3, 4))
    torch.testing.assert_close(e, s.normalize_embeddings(emb(ids)))
    assert s.debug_str() == "local-l2-eps1e-06"
    ib = AL.InBatchNegativesSampler(l2_norm=False, l2_norm_eps=1e-6, dedup_embeddings=True)
    x = torch.randn(2, 3, 8)
    ib.process_batch(torch.tensor([[1, 2, 2], [3, 1, 0]]), torch.tensor([[True, True, True], [True, True, False]]), x)
    cids, cemb = ib.get_all_ids_and_embeddings()
    assert sorted(cids.tolist()) == [1, 2, 3] and cemb.shape == (3, 8)
    i2, e2 = ib(positive_ids=torch.tensor([1, 2]), num_to_sample=5)
    assert i2.shape == (2, 5) and e2.shape == (2, 5, 8) and set(i2.flatten().tolist()) <= {1, 2, 3}
    assert ib.debug_str() == "in-batch-dedup"
    loss = SS.SampledSoftmaxLoss(num_to_sample=4, softmax_temperature=0.05)
    with pytest.raises(RuntimeError, match="GPU"):
        loss.jagged_forward(torch.randn(3, 8), torch.tensor([1, 2, 3]), torch.randn(3, 8), torch.ones(3), s)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/hstu_hip.h is the FFI contract: it must compile as strict C99 (cgo / JNI / N-API style bindings parse
    it as C) and a C program must link against the library and call it (no GPU needed for hstu_abi_version)."""
    import shutil
    import subprocess
    from generative_recommenders_amd import _lib

    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists(_lib.LIB_PATH):
        pytest.skip("gcc or the built library is not available")
    src = tmp_path / "abi_c_check.c"
    src.write_text('#include <stdio.h>\n#include "hstu_hip.h"\n'
                   "int main(void) {\n  HstuAttnParams p; HstuAttnBwdParams bp; (void)p; (void)bp;\n"
                   '  printf("%d\\n", hstu_abi_version());\n  return hstu_abi_version() == HSTU_ABI_VERSION ? 0 : 1;\n}\n')
    exe = tmp_path / "abi_c_check"
    libdir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                        str(exe), "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + libdir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and out.stdout.strip() == str(_lib.ABI_VERSION), (out.stdout, out.stderr)


def test_compiled_operator_library_registers_schemas_and_meta_kernels():
    """libhstu_torch_ops.so: torch.ops.load_library() alone yields the reference's hstu:: schemas (argument for argument:
    flash_api.cpp:275-352, cpp_ops.cpp:94-102) with Meta kernels for every operator (shape inference on the CPU box);
    CPU tensors are refused by the dispatcher (no CPU kernels, no fallback)."""
    import pytest
    import torch

    from generative_recommenders_amd.ops import torch_library

    torch_library.register()
    torch_library.register()
    s = str(torch.ops.hstu.hstu_mha_bwd.default._schema)
    assert s == ("hstu::hstu_mha_bwd(int max_seq_len, float alpha, Tensor dout, Tensor q, Tensor k, Tensor v, Tensor dq, Tensor dk, "
                 "Tensor dv, Tensor? seq_offsets, bool causal, Tensor? num_targets, Tensor? attn_scale, int max_attn_len, "
                 "int min_full_attn_seq_len, int contextual_seq_len, bool sort_by_length, bool deterministic, int sm_margin) -> Tensor[]")
    assert "SymInt max_seq_len, float alpha, Tensor q, Tensor k, Tensor v, Tensor? seq_offsets, bool causal" in str(
        torch.ops.hstu.hstu_mha.default._schema)
    m = lambda *shape, dtype=torch.bfloat16: torch.empty(*shape, dtype=dtype, device="meta")
    q, v, off = m(10, 2, 16), m(10, 2, 24), m(3, dtype=torch.int64)
    assert torch.ops.hstu.hstu_mha(20, 0.25, q, q, v, off, True, None, None, 0, 0, 0, None, None, None, False, False, 0).shape == (10, 2, 24)
    assert torch.ops.hstu.hstu_mha_fwd(20, 0.25, m(3, 20, 2, 16), m(3, 20, 2, 16), m(3, 20, 2, 24), None, True, None, None, 0, 0, 0,
                                       None, None, None, 0).shape == (3, 20, 2, 24)
    grads = torch.ops.hstu.hstu_mha_bwd(20, 0.25, v, q, q, v, m(10, 2, 16), m(10, 2, 16), m(10, 2, 24), off, True, None, None, 0, 0, 0,
                                        False, False, 0)
    assert [tuple(g.shape) for g in grads] == [(10, 2, 16), (10, 2, 16), (10, 2, 24)]
    i64 = lambda n: m(n, dtype=torch.int64)
    assert torch.ops.hstu.complete_cumsum(i64(5)).shape == (6,)
    assert torch.ops.hstu.expand_1d_jagged_to_dense(i64(7), off, 4).shape == (2, 4)
    assert torch.ops.hstu.concat_1d_jagged_jagged(i64(2), i64(7), i64(2), i64(5)).shape == (12,)
    assert [tuple(t.shape) for t in torch.ops.hstu.sort_kv_pairs(i64(6), i64(6))] == [(6,), (6,)]
    with pytest.raises((RuntimeError, NotImplementedError), match="CPU"):
        torch.ops.hstu.complete_cumsum(torch.arange(4))
