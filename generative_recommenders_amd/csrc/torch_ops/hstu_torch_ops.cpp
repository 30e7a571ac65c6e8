// torch.library seam of the HSTU HIP kernels: a compiled library that registers the reference's `hstu::` operator
// schemas -- argument for argument -- with CUDA (= HIP tensors under PyTorch-ROCm) and Meta kernels, so that
//     torch.ops.load_library(".../libhstu_torch_ops.so")
// is all a caller of torch.ops.hstu.* needs, exactly as with the reference's extension
// (ops/cpp/cuda_hstu_attention.py:21-23, hstu_attention/flash_api.cpp:275-365, flash_meta.cpp, cpp_ops.cpp:94-135).
// Host-side C++ only: every op marshals tensors into the C ABI of libhstu_hip.so (include/hstu_hip.h) and launches on
// torch's current HIP stream.  No CPU kernels are registered: the reference's are dummies that return empty tensors
// (flash_cpu_dummy.cpp); here a CPU tensor fails in the dispatcher.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // PyTorch-ROCm tensors carry DeviceType::CUDA
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <optional>
#include <tuple>
#include <vector>

#include "../../../include/hstu_hip.h"

namespace hstu_ops {

using at::Tensor;
using OptT = std::optional<Tensor>;

static void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, ": libhstu_hip error ", rc, ": ", hstu_last_error()); }
static void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }
static int dtype_code(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kBFloat16: return HSTU_DTYPE_BF16;
    case at::kHalf: return HSTU_DTYPE_F16;
    case at::kFloat: return HSTU_DTYPE_F32;
    default: TORCH_CHECK(false, "HSTU HIP ops support bf16 / fp16 / fp32 tensors, got ", t.scalar_type());
  }
}
static Tensor index_tensor(const Tensor& t) {
  Tensor r = (t.scalar_type() == at::kInt || t.scalar_type() == at::kLong) ? t : t.to(at::kLong);
  return r.contiguous();
}
static int index_code(const Tensor& t) { return t.scalar_type() == at::kLong ? HSTU_INDEX_I64 : HSTU_INDEX_I32; }

// head dims that are not a multiple of the 16-byte vector are zero-padded (zeros change neither q.k nor the sliced output)
static Tensor pad_head_dim(const Tensor& t) {
  const int64_t mult = 16 / t.element_size(), pad = (mult - t.size(-1) % mult) % mult;
  return pad ? at::constant_pad_nd(t, {0, pad}) : t;
}
// (rows, H, d) with contiguous last dim and 16-byte aligned (row, head) vectors; copies only when the layout forces it
static Tensor aligned_rows(const Tensor& t) {
  const int64_t es = t.element_size();
  const bool ok = t.stride(-1) == 1 && (t.stride(0) * es) % 16 == 0 && (t.stride(1) * es) % 16 == 0 && ((uintptr_t)t.data_ptr() % 16) == 0;
  return ok ? t : t.contiguous();
}

struct Jagged {       // the (total rows, H, d) view of q / k / v plus offsets: dense (B, S, H, d) inputs get arange(B + 1) * S
  Tensor q, k, v, offsets;
  bool dense;
  int64_t B, S;
};
static Jagged as_jagged(const Tensor& q, const Tensor& k, const Tensor& v, const OptT& seq_offsets, int64_t max_seq_len) {
  Jagged j;
  j.dense = !seq_offsets.has_value();
  if (!j.dense) {
    TORCH_CHECK(q.dim() == 3 && k.dim() == 3 && v.dim() == 3, "jagged q, k, v must be (total rows, heads, dim)");
    j.q = q; j.k = k; j.v = v;
    j.offsets = index_tensor(*seq_offsets);
    j.B = j.offsets.numel() - 1; j.S = max_seq_len;
    return j;
  }
  TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.dim() == 4, "dense q, k, v must be (batch, seq, heads, dim)");
  j.B = q.size(0); j.S = q.size(1);
  TORCH_CHECK(j.S == max_seq_len, "dense input: max_seq_len must equal the sequence dimension");
  j.q = q.reshape({j.B * j.S, q.size(2), q.size(3)});
  j.k = k.reshape({j.B * j.S, k.size(2), k.size(3)});
  j.v = v.reshape({j.B * j.S, v.size(2), v.size(3)});
  j.offsets = at::arange(j.B + 1, q.options().dtype(at::kLong)) * j.S;
  return j;
}

// The parameter structs cross into libhstu_hip.so by pointer: a core library of another ABI version would read other
// fields at these offsets.  Checked once per process, on the first call that fills a struct.
static void check_core_abi() {
  static const int got = hstu_abi_version();
  TORCH_CHECK(got == HSTU_ABI_VERSION, "libhstu_torch_ops.so was built against libhstu_hip ABI v", HSTU_ABI_VERSION,
              " but the loaded libhstu_hip.so reports v", got, ": rebuild both (generative_recommenders_amd._lib.build())");
}

static void fill(HstuAttnParams& p, const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& offsets, const OptT& num_targets,
                 Tensor& nt_keep, int64_t max_seq_len, double alpha, const OptT& attn_scale, Tensor& scale_keep,
                 int64_t max_attn_len, int64_t min_full, int64_t contextual) {
  check_core_abi();
  memset(&p, 0, sizeof(p));
  p.q = q.data_ptr(); p.k = k.data_ptr(); p.v = v.data_ptr();
  p.seq_offsets = offsets.data_ptr();
  p.q_row_stride = q.stride(0); p.q_head_stride = q.stride(1);
  p.k_row_stride = k.stride(0); p.k_head_stride = k.stride(1);
  p.v_row_stride = v.stride(0); p.v_head_stride = v.stride(1);
  p.batch = (int32_t)(offsets.numel() - 1);
  p.heads = (int32_t)q.size(1); p.dqk = (int32_t)q.size(2); p.dv = (int32_t)v.size(2);
  p.max_seq_len = (int32_t)max_seq_len;
  p.alpha = (float)alpha; p.scale = 1.0f / (float)max_seq_len;
  p.max_attn_len = (int32_t)max_attn_len; p.contextual_seq_len = (int32_t)contextual; p.min_full_attn_seq_len = (int32_t)min_full;
  p.dtype = dtype_code(q);
  p.offsets_dtype = index_code(offsets);
  if (num_targets.has_value()) {
    nt_keep = index_tensor(*num_targets);
    p.num_targets = nt_keep.data_ptr();
    p.targets_dtype = index_code(nt_keep);
  }
  if (attn_scale.has_value()) {   // element 0 replaces 1/N, read on the device (flash_api.cpp:283, mainloop_fwd_sm80.h:790-793)
    scale_keep = attn_scale->to(at::kFloat).contiguous();
    TORCH_CHECK(scale_keep.is_cuda() && scale_keep.numel() >= 1, "attn_scale must be a non-empty GPU tensor");
    p.attn_scale = (const float*)scale_keep.data_ptr();
  }
}

static void reject_fp8(const OptT& a, const OptT& b, const OptT& c) {
  TORCH_CHECK(!a.has_value() && !b.has_value() && !c.has_value(), "hstu_mha: fp8 descale tensors are not supported on gfx950 (no fp8 instantiation)");
}

Tensor hstu_mha_fwd(const at::SymInt max_seq_len_s, double alpha, Tensor& q, Tensor& k, Tensor& v, const OptT& seq_offsets, bool causal,
                    const OptT& num_targets, const OptT& attn_scale, int64_t max_attn_len, int64_t min_full_attn_seq_len,
                    int64_t contextual_seq_len, const OptT& q_descale, const OptT& k_descale, const OptT& v_descale,
                    const int64_t sm_margin) {
  reject_fp8(q_descale, k_descale, v_descale);
  TORCH_CHECK(causal, "only support causal attention");
  const int64_t N = max_seq_len_s.expect_int();
  TORCH_CHECK(N > 0, "max_seq_len must be larger than 0");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(q.device());
  Jagged j = as_jagged(q, k, v, seq_offsets, N);
  const int64_t dv = j.v.size(2);
  Tensor qp = aligned_rows(pad_head_dim(j.q)), kp = aligned_rows(pad_head_dim(j.k)), vp = aligned_rows(pad_head_dim(j.v));
  Tensor out = at::empty({qp.size(0), qp.size(1), vp.size(2)}, qp.options());
  if (qp.size(0) > 0) {
    HstuAttnParams p;
    Tensor nt_keep, scale_keep;
    fill(p, qp, kp, vp, j.offsets, num_targets, nt_keep, N, alpha, attn_scale, scale_keep, max_attn_len, min_full_attn_seq_len, contextual_seq_len);
    p.out = out.data_ptr(); p.o_row_stride = out.stride(0); p.o_head_stride = out.stride(1);
    check(hstu_attn_fwd(&p, stream_of(qp)), "hstu_mha_fwd");
  }
  if (out.size(2) != dv) out = out.slice(2, 0, dv).contiguous();
  return j.dense ? out.reshape({j.B, j.S, out.size(1), out.size(2)}) : out;
}

std::vector<Tensor> hstu_mha_bwd(int64_t max_seq_len, double alpha, Tensor& dout, Tensor& q, Tensor& k, Tensor& v, Tensor& dq, Tensor& dk,
                                 Tensor& dv, const OptT& seq_offsets, bool causal, const OptT& num_targets, const OptT& attn_scale,
                                 int64_t max_attn_len, int64_t min_full_attn_seq_len, int64_t contextual_seq_len, bool sort_by_length,
                                 bool deterministic, const int64_t sm_margin) {
  TORCH_CHECK(causal, "only support causal attention");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(q.device());
  Jagged j = as_jagged(q, k, v, seq_offsets, max_seq_len);
  Tensor d_o = j.dense ? dout.reshape({j.B * j.S, dout.size(2), dout.size(3)}) : dout;
  Tensor dq3 = j.dense ? dq.view({j.B * j.S, dq.size(2), dq.size(3)}) : dq;
  Tensor dk3 = j.dense ? dk.view({j.B * j.S, dk.size(2), dk.size(3)}) : dk;
  Tensor dv3 = j.dense ? dv.view({j.B * j.S, dv.size(2), dv.size(3)}) : dv;
  const int64_t es = q.element_size();
  const bool padded = (j.q.size(2) * es) % 16 || (j.v.size(2) * es) % 16;
  Tensor qp = aligned_rows(pad_head_dim(j.q)), kp = aligned_rows(pad_head_dim(j.k)), vp = aligned_rows(pad_head_dim(j.v));
  Tensor dop = aligned_rows(pad_head_dim(d_o));
  auto writable = [&](const Tensor& t) { return !padded && aligned_rows(t).is_same(t); };
  // gradients go straight into the caller's (possibly strided) dq / dk / dv when their layout allows it, as the reference does
  Tensor gq = writable(dq3) ? dq3 : at::empty_like(qp), gk = writable(dk3) ? dk3 : at::empty_like(kp), gv = writable(dv3) ? dv3 : at::empty_like(vp);
  if (qp.size(0) > 0) {
    HstuAttnBwdParams bp;
    memset(&bp, 0, sizeof(bp));
    Tensor nt_keep, scale_keep;
    fill(bp.fwd, qp, kp, vp, j.offsets, num_targets, nt_keep, max_seq_len, alpha, attn_scale, scale_keep, max_attn_len, min_full_attn_seq_len,
         contextual_seq_len);
    bp.dout = dop.data_ptr(); bp.dq = gq.data_ptr(); bp.dk = gk.data_ptr(); bp.dv = gv.data_ptr();
    bp.do_row_stride = dop.stride(0); bp.do_head_stride = dop.stride(1);
    bp.dq_row_stride = gq.stride(0); bp.dq_head_stride = gq.stride(1);
    bp.dk_row_stride = gk.stride(0); bp.dk_head_stride = gk.stride(1);
    bp.dv_row_stride = gv.stride(0); bp.dv_head_stride = gv.stride(1);
    bp.total_rows = qp.size(0);
    // deterministic (flash_api.cpp:291): a sequence that fits ONE key block (max_seq_len <= 224 at 128-wide 16-bit heads) has every
    // sum in a fixed order anyway.  Longer sequences: each key block's fp32 dq partial goes to a slab of its own and the slabs are
    // added in block order (ABI v8; the CUDA reference serialises its adds with a semaphore, flash_common.cpp:806-858) -- the
    // workspace grows by the number of key blocks.
    bp.deterministic = deterministic ? 1 : 0;
    Tensor ws;
    const size_t ws_bytes = hstu_attn_bwd_workspace_bytes(&bp);
    if (ws_bytes) {
      ws = at::empty({(int64_t)ws_bytes}, qp.options().dtype(at::kByte));
      bp.workspace = ws.data_ptr();
    }
    check(hstu_attn_bwd(&bp, stream_of(qp)), "hstu_mha_bwd");
  }
  if (!gq.is_same(dq3)) dq3.copy_(gq.slice(2, 0, dq3.size(2)));
  if (!gk.is_same(dk3)) dk3.copy_(gk.slice(2, 0, dk3.size(2)));
  if (!gv.is_same(dv3)) dv3.copy_(gv.slice(2, 0, dv3.size(2)));
  return {dq, dk, dv};
}

// hstu_mha = the autograd node over the two ops above (HSTUFlashAttentionFunction, flash_api.cpp:34-160)
class HstuMhaFunction : public torch::autograd::Function<HstuMhaFunction> {
 public:
  static Tensor forward(torch::autograd::AutogradContext* ctx, const at::SymInt max_seq_len, double alpha, Tensor q, Tensor k, Tensor v,
                        const OptT& seq_offsets, bool causal, const OptT& num_targets, const OptT& attn_scale, int64_t max_attn_len,
                        int64_t min_full_attn_seq_len, int64_t contextual_seq_len, const OptT& q_descale, const OptT& k_descale,
                        const OptT& v_descale, bool sort_by_length, bool deterministic, int64_t sm_margin) {
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("hstu::hstu_mha_fwd", "").typed<decltype(hstu_mha_fwd)>();
    ctx->save_for_backward({q, k, v, seq_offsets.value_or(Tensor()), num_targets.value_or(Tensor()), attn_scale.value_or(Tensor())});
    ctx->saved_data["max_seq_len"] = max_seq_len.expect_int();
    ctx->saved_data["alpha"] = alpha;
    ctx->saved_data["causal"] = causal;
    ctx->saved_data["max_attn_len"] = max_attn_len;
    ctx->saved_data["min_full_attn_seq_len"] = min_full_attn_seq_len;
    ctx->saved_data["contextual_seq_len"] = contextual_seq_len;
    ctx->saved_data["sort_by_length"] = sort_by_length;
    ctx->saved_data["deterministic"] = deterministic;
    ctx->saved_data["sm_margin"] = sm_margin;
    at::AutoDispatchBelowADInplaceOrView below;
    return op.call(max_seq_len, alpha, q, k, v, seq_offsets, causal, num_targets, attn_scale, max_attn_len, min_full_attn_seq_len,
                   contextual_seq_len, q_descale, k_descale, v_descale, sm_margin);
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("hstu::hstu_mha_bwd", "").typed<decltype(hstu_mha_bwd)>();
    auto saved = ctx->get_saved_variables();
    Tensor q = saved[0], k = saved[1], v = saved[2];
    auto opt = [](const Tensor& t) { return t.defined() ? OptT(t) : std::nullopt; };
    Tensor dout = grads[0].contiguous();
    Tensor dq = at::empty_like(q), dk = at::empty_like(k), dv = at::empty_like(v);
    op.call(ctx->saved_data["max_seq_len"].toInt(), ctx->saved_data["alpha"].toDouble(), dout, q, k, v, dq, dk, dv, opt(saved[3]),
            ctx->saved_data["causal"].toBool(), opt(saved[4]), opt(saved[5]), ctx->saved_data["max_attn_len"].toInt(),
            ctx->saved_data["min_full_attn_seq_len"].toInt(), ctx->saved_data["contextual_seq_len"].toInt(),
            ctx->saved_data["sort_by_length"].toBool(), ctx->saved_data["deterministic"].toBool(), ctx->saved_data["sm_margin"].toInt());
    torch::autograd::variable_list out(18);
    out[2] = dq; out[3] = dk; out[4] = dv;
    return out;
  }
};

// below autograd (inference / no_grad calls land here directly): the forward op
Tensor hstu_mha_cuda(const at::SymInt max_seq_len, double alpha, const Tensor& q, const Tensor& k, const Tensor& v, const OptT& seq_offsets, bool causal,
                     const OptT& num_targets, const OptT& attn_scale, int64_t max_attn_len, int64_t min_full_attn_seq_len,
                     int64_t contextual_seq_len, const OptT& q_descale, const OptT& k_descale, const OptT& v_descale, bool sort_by_length,
                     bool deterministic, int64_t sm_margin) {
  Tensor qq = q, kk = k, vv = v;
  return hstu_mha_fwd(max_seq_len, alpha, qq, kk, vv, seq_offsets, causal, num_targets, attn_scale, max_attn_len, min_full_attn_seq_len,
                      contextual_seq_len, q_descale, k_descale, v_descale, sm_margin);
}

Tensor hstu_mha(const at::SymInt max_seq_len, double alpha, const Tensor& q, const Tensor& k, const Tensor& v, const OptT& seq_offsets, bool causal,
                const OptT& num_targets, const OptT& attn_scale, int64_t max_attn_len, int64_t min_full_attn_seq_len,
                int64_t contextual_seq_len, const OptT& q_descale, const OptT& k_descale, const OptT& v_descale, bool sort_by_length,
                bool deterministic, int64_t sm_margin) {
  return HstuMhaFunction::apply(max_seq_len, alpha, q, k, v, seq_offsets, causal, num_targets, attn_scale, max_attn_len,
                                min_full_attn_seq_len, contextual_seq_len, q_descale, k_descale, v_descale, sort_by_length, deterministic,
                                sm_margin);
}

// ---- Meta kernels (shapes only; flash_meta.cpp)
static Tensor fwd_out_meta(const at::SymInt& max_seq_len, const Tensor& q, const Tensor& v, const OptT& seq_offsets) {
  auto qs = q.sym_sizes();
  auto vd = v.sym_sizes().back();
  if (seq_offsets.has_value()) return at::empty_symint({qs[0], qs[1], vd}, q.options());
  return at::empty_symint({qs[0], max_seq_len, qs[2], vd}, q.options());
}
Tensor hstu_mha_fwd_meta(const at::SymInt max_seq_len, double, Tensor& q, Tensor&, Tensor& v, const OptT& seq_offsets, bool, const OptT&, const OptT&,
                         int64_t, int64_t, int64_t, const OptT&, const OptT&, const OptT&, const int64_t) {
  return fwd_out_meta(max_seq_len, q, v, seq_offsets);
}
Tensor hstu_mha_meta(const at::SymInt max_seq_len, double, const Tensor& q, const Tensor&, const Tensor& v, const OptT& seq_offsets, bool, const OptT&,
                     const OptT&, int64_t, int64_t, int64_t, const OptT&, const OptT&, const OptT&, bool, bool, int64_t) {
  return fwd_out_meta(max_seq_len, q, v, seq_offsets);
}
std::vector<Tensor> hstu_mha_bwd_meta(int64_t, double, Tensor&, Tensor&, Tensor&, Tensor&, Tensor& dq, Tensor& dk, Tensor& dv, const OptT&, bool, const OptT&,
                                      const OptT&, int64_t, int64_t, int64_t, bool, bool, const int64_t) {
  return {dq, dk, dv};
}

// ---- jagged helpers (cpp_ops.cpp:94-135)
Tensor complete_cumsum(const Tensor& values) {
  TORCH_CHECK(values.dim() == 1, "complete_cumsum: values must be 1-D");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(values.device());
  Tensor in = index_tensor(values);
  Tensor out = at::empty({in.numel() + 1}, in.options());
  check(hstu_complete_cumsum(in.data_ptr(), out.data_ptr(), in.numel(), index_code(in), stream_of(in)), "complete_cumsum");
  return out;
}
Tensor complete_cumsum_meta(const Tensor& values) { return at::empty_symint({values.sym_numel() + 1}, values.options()); }

Tensor expand_1d_jagged_to_dense(const Tensor& values, const Tensor& offsets, const at::SymInt max_len_s) {
  const int64_t max_len = max_len_s.expect_int();
  TORCH_CHECK(values.element_size() == 4 || values.element_size() == 8, "expand_1d_jagged_to_dense: 4- or 8-byte elements");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(values.device());
  Tensor vals = values.contiguous(), off = index_tensor(offsets);
  const int64_t B = off.numel() - 1;
  Tensor out = at::empty({B, max_len}, vals.options());
  if (out.numel())
    check(hstu_expand_1d_jagged_to_dense(vals.data_ptr(), off.data_ptr(), out.data_ptr(), (int32_t)B, (int32_t)max_len,
                                         (int32_t)vals.element_size(), index_code(off), stream_of(vals)), "expand_1d_jagged_to_dense");
  return out;
}
Tensor expand_1d_jagged_to_dense_meta(const Tensor& values, const Tensor& offsets, const at::SymInt max_len) {
  return at::empty_symint({offsets.sym_numel() - 1, max_len}, values.options());
}

Tensor concat_1d_jagged_jagged(const Tensor& lengths_left, const Tensor& values_left, const Tensor& lengths_right, const Tensor& values_right) {
  TORCH_CHECK(values_left.scalar_type() == values_right.scalar_type() && (values_left.element_size() == 4 || values_left.element_size() == 8),
              "concat_1d_jagged_jagged: values of one 4- or 8-byte dtype");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(values_left.device());
  Tensor ol = complete_cumsum(lengths_left.to(at::kLong)), orr = complete_cumsum(lengths_right.to(at::kLong));
  Tensor vl = values_left.contiguous(), vr = values_right.contiguous();
  Tensor out = at::empty({vl.numel() + vr.numel()}, vl.options());
  if (out.numel())
    check(hstu_concat_1d_jagged_jagged(vl.data_ptr(), ol.data_ptr(), vr.data_ptr(), orr.data_ptr(), out.data_ptr(),
                                       (int32_t)lengths_left.numel(), (int32_t)vl.element_size(), HSTU_INDEX_I64, stream_of(vl)),
          "concat_1d_jagged_jagged");
  return out;
}
Tensor concat_1d_jagged_jagged_meta(const Tensor&, const Tensor& values_left, const Tensor&, const Tensor& values_right) {
  return at::empty_symint({values_left.sym_numel() + values_right.sym_numel()}, values_left.options());
}

// stable radix sort of (key, value) pairs on key bits [0, end_bit) (sort_kv_pairs_cuda.cpp); the sort itself is torch's
// (rocPRIM on the GPU), as the reference's is cub's: index plumbing, not the hot path
std::tuple<Tensor, Tensor> sort_kv_pairs(const Tensor& keys, const Tensor& values, const std::optional<int64_t>& end_bit, bool descending) {
  TORCH_CHECK(keys.dim() == 1 && values.dim() == 1 && keys.sizes() == values.sizes(), "sort_kv_pairs: keys and values must be 1-D tensors of one length");
  const auto kt = keys.scalar_type();
  TORCH_CHECK(kt == at::kInt || kt == at::kLong || kt == at::kByte || kt == at::kShort, "sort_kv_pairs: keys must be int32, int64, uint8 or int16");
  const int64_t width = keys.element_size() * 8;
  Tensor sub = keys;
  if (end_bit.has_value() && *end_bit < width) {
    if (*end_bit <= 0) return {keys.clone(), values.clone()};
    sub = at::bitwise_and(keys.to(at::kLong), (int64_t)((1LL << *end_bit) - 1));
  }
  Tensor order = std::get<1>(at::sort(sub, /*stable=*/true, /*dim=*/0, descending));
  return {keys.index_select(0, order), values.index_select(0, order)};
}
std::tuple<Tensor, Tensor> sort_kv_pairs_meta(const Tensor& keys, const Tensor& values, const std::optional<int64_t>&, bool) {
  return {at::empty_like(keys), at::empty_like(values)};
}

}  // namespace hstu_ops

TORCH_LIBRARY_FRAGMENT(hstu, m) {
  m.def("hstu_mha(SymInt max_seq_len, float alpha, Tensor q, Tensor k, Tensor v, Tensor? seq_offsets, bool causal, Tensor? num_targets, "
        "Tensor? attn_scale, int max_attn_len, int min_full_attn_seq_len, int contextual_seq_len, Tensor? q_descale, Tensor? k_descale, "
        "Tensor? v_descale, bool sort_by_length, bool deterministic, int sm_margin) -> Tensor");
  m.def("hstu_mha_fwd(SymInt max_seq_len, float alpha, Tensor q, Tensor k, Tensor v, Tensor? seq_offsets, bool causal, Tensor? num_targets, "
        "Tensor? attn_scale, int max_attn_len, int min_full_attn_seq_len, int contextual_seq_len, Tensor? q_descale, Tensor? k_descale, "
        "Tensor? v_descale, int sm_margin) -> Tensor");
  m.def("hstu_mha_bwd(int max_seq_len, float alpha, Tensor dout, Tensor q, Tensor k, Tensor v, Tensor dq, Tensor dk, Tensor dv, "
        "Tensor? seq_offsets, bool causal, Tensor? num_targets, Tensor? attn_scale, int max_attn_len, int min_full_attn_seq_len, "
        "int contextual_seq_len, bool sort_by_length,bool deterministic,int sm_margin) -> Tensor[]");
  m.def("expand_1d_jagged_to_dense(Tensor values, Tensor offsets, SymInt max_len) -> Tensor");
  m.def("concat_1d_jagged_jagged(Tensor lengths_left, Tensor values_left, Tensor lengths_right, Tensor values_right) -> Tensor");
  m.def("complete_cumsum(Tensor values) -> Tensor");
  m.def("sort_kv_pairs(Tensor keys, Tensor values, int? end_bit=None, bool descending=False) -> (Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(hstu, CUDA, m) {
  m.impl("hstu_mha", hstu_ops::hstu_mha_cuda);
  m.impl("hstu_mha_fwd", hstu_ops::hstu_mha_fwd);
  m.impl("hstu_mha_bwd", hstu_ops::hstu_mha_bwd);
  m.impl("expand_1d_jagged_to_dense", hstu_ops::expand_1d_jagged_to_dense);
  m.impl("concat_1d_jagged_jagged", hstu_ops::concat_1d_jagged_jagged);
  m.impl("complete_cumsum", hstu_ops::complete_cumsum);
}

// index plumbing on at::sort: runs on whatever device the tensors live on
TORCH_LIBRARY_IMPL(hstu, CompositeExplicitAutograd, m) { m.impl("sort_kv_pairs", hstu_ops::sort_kv_pairs); }

TORCH_LIBRARY_IMPL(hstu, Meta, m) {
  m.impl("hstu_mha", hstu_ops::hstu_mha_meta);
  m.impl("hstu_mha_fwd", hstu_ops::hstu_mha_fwd_meta);
  m.impl("hstu_mha_bwd", hstu_ops::hstu_mha_bwd_meta);
  m.impl("expand_1d_jagged_to_dense", hstu_ops::expand_1d_jagged_to_dense_meta);
  m.impl("concat_1d_jagged_jagged", hstu_ops::concat_1d_jagged_jagged_meta);
  m.impl("complete_cumsum", hstu_ops::complete_cumsum_meta);
  m.impl("sort_kv_pairs", hstu_ops::sort_kv_pairs_meta);
}

TORCH_LIBRARY_IMPL(hstu, Autograd, m) {
  m.impl("hstu_mha", hstu_ops::hstu_mha);    // the autograd node (HSTUFlashAttentionFunction in the reference, flash_api.cpp:34-160)
  m.impl("expand_1d_jagged_to_dense", torch::autograd::autogradNotImplementedFallback());
  m.impl("complete_cumsum", torch::autograd::autogradNotImplementedFallback());
}
