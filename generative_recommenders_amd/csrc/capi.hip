// C ABI entry points of libhstu_hip.so (see include/hstu_hip.h): argument validation in the
// spirit of the reference's TORCH_CHECKs (ops/cpp/hstu_attention/flash_common.cpp:339-456),
// then dtype dispatch to the per-dtype launchers.
#include <stdarg.h>
#include <stdio.h>

#include "capi_internal.h"

namespace hstu {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "%s: HIP launch failed: %s", what, hipGetErrorString(e));
  return HSTU_OK;
}

static int validate_attn(const HstuAttnParams& p, const char* who) {
  if (!p.q || !p.k || !p.v || !p.seq_offsets) return set_error(HSTU_EINVAL, "%s: q, k, v and seq_offsets must be non-NULL", who);
  if (p.batch < 0 || p.heads <= 0) return set_error(HSTU_EINVAL, "%s: bad batch/heads", who);
  if (p.max_seq_len <= 0) return set_error(HSTU_EINVAL, "%s: max_seq_len must be larger than 0", who);
  if (p.dtype != HSTU_DTYPE_BF16 && p.dtype != HSTU_DTYPE_F16 && p.dtype != HSTU_DTYPE_F32)
    return set_error(HSTU_EINVAL, "%s: dtype must be bf16, fp16 or fp32", who);
  const int eb = p.dtype == HSTU_DTYPE_F32 ? 4 : 2;
  const int epu = 16 / eb;
  if (p.dqk <= 0 || p.dv <= 0 || p.dqk % epu || p.dv % epu)
    return set_error(HSTU_EINVAL, "%s: head dims (%d, %d) must be positive multiples of %d", who, p.dqk, p.dv, epu);
  if (!pad_head_dim(p.dqk) || !pad_head_dim(p.dv))
    return set_error(HSTU_EUNSUPPORTED, "%s: head dims (%d, %d) above 128 are not instantiated", who, p.dqk, p.dv);
  const int64_t strides[] = {p.q_row_stride, p.q_head_stride, p.k_row_stride, p.k_head_stride, p.v_row_stride, p.v_head_stride};
  for (int64_t s : strides)
    if ((s * eb) % 16) return set_error(HSTU_EINVAL, "%s: every (row, head) vector must be 16-byte aligned (stride %lld elements)", who, (long long)s);
  if (((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v) & 15) return set_error(HSTU_EINVAL, "%s: q/k/v base pointers must be 16-byte aligned", who);
  if (p.max_attn_len < 0 || p.contextual_seq_len < 0 || p.min_full_attn_seq_len < 0)
    return set_error(HSTU_EINVAL, "%s: negative mask parameter", who);
  if (p.delta_q < 0) return set_error(HSTU_EINVAL, "%s: negative delta_q", who);
  if (p.pos_w) {
    if ((p.ts_w == nullptr) != (p.timestamps == nullptr)) return set_error(HSTU_EINVAL, "%s: ts_w and timestamps must be given together", who);
    if (p.ts_w && (p.num_buckets <= 0 || !(p.bucket_div > 0.f) || p.ts_row_stride < p.max_seq_len))
      return set_error(HSTU_EINVAL, "%s: bad bucket parameters / timestamp stride", who);
  }
  return HSTU_OK;
}

}  // namespace hstu

using namespace hstu;

extern "C" {

int hstu_abi_version(void) { return HSTU_ABI_VERSION; }
const char* hstu_last_error(void) { return g_err; }

int hstu_attn_fwd(const HstuAttnParams* p, void* stream) {
  if (!p) return set_error(HSTU_EINVAL, "hstu_attn_fwd: NULL params");
  if (int e = validate_attn(*p, "hstu_attn_fwd")) return e;
  if (!p->out) return set_error(HSTU_EINVAL, "hstu_attn_fwd: out is NULL");
  if (((p->o_row_stride | p->o_head_stride) * (p->dtype == HSTU_DTYPE_F32 ? 4 : 2)) % 16 || ((uintptr_t)p->out & 15))
    return set_error(HSTU_EINVAL, "hstu_attn_fwd: out rows must be 16-byte aligned");
  if (p->batch == 0) return HSTU_OK;   // empty batch: nothing to launch (flash_common.cpp:548-551)
  hipStream_t st = (hipStream_t)stream;
  switch (p->dtype) {
    case HSTU_DTYPE_BF16: return launch_attn_fwd_bf16(*p, st);
    case HSTU_DTYPE_F16: return launch_attn_fwd_f16(*p, st);
    default: return launch_attn_fwd_f32(*p, st);
  }
}

int hstu_attn_fwd_kernel_name(const HstuAttnParams* p, char* buf, size_t len) {
  if (!p) return set_error(HSTU_EINVAL, "hstu_attn_fwd_kernel_name: NULL params");
  return attn_kernel_name(*p, nullptr, buf, len);
}

int hstu_attn_bwd_kernel_name(const HstuAttnBwdParams* p, char* buf, size_t len) {
  if (!p) return set_error(HSTU_EINVAL, "hstu_attn_bwd_kernel_name: NULL params");
  return attn_kernel_name(p->fwd, p, buf, len);
}

size_t hstu_attn_bwd_workspace_bytes(const HstuAttnBwdParams* p) {
  if (!p) return 0;
  return attn_bwd_workspace_bytes(*p);
}

int hstu_attn_bwd(const HstuAttnBwdParams* p, void* stream) {
  if (!p) return set_error(HSTU_EINVAL, "hstu_attn_bwd: NULL params");
  if (int e = validate_attn(p->fwd, "hstu_attn_bwd")) return e;
  if (p->fwd.delta_q != 0) return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd: delta_q attention is forward-only (as in the reference)");
  if (!p->dout || !p->dq || !p->dk || !p->dv) return set_error(HSTU_EINVAL, "hstu_attn_bwd: dout, dq, dk, dv must be non-NULL");
  const int eb = p->fwd.dtype == HSTU_DTYPE_F32 ? 4 : 2;
  const int64_t strides[] = {p->do_row_stride, p->do_head_stride, p->dq_row_stride, p->dq_head_stride,
                             p->dk_row_stride, p->dk_head_stride, p->dv_row_stride, p->dv_head_stride};
  for (int64_t s : strides)
    if ((s * eb) % 16) return set_error(HSTU_EINVAL, "hstu_attn_bwd: gradient rows must be 16-byte aligned (stride %lld elements)", (long long)s);
  if (((uintptr_t)p->dout | (uintptr_t)p->dq | (uintptr_t)p->dk | (uintptr_t)p->dv) & 15)
    return set_error(HSTU_EINVAL, "hstu_attn_bwd: gradient base pointers must be 16-byte aligned");
  if (p->fwd.pos_w && (!p->dpos_w || (p->fwd.ts_w && !p->dts_w)))
    return set_error(HSTU_EINVAL, "hstu_attn_bwd: dpos_w / dts_w outputs are required with a relative bias");
  // ahead of every dispatch decision (solo_bias, fold_bias and the general bias kernel all add their table gradients with
  // float atomics into LDS histograms): refused rather than silently not honoured
  if (p->fwd.pos_w && p->deterministic)
    return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd: deterministic = 1 is not available with the relative bias (the table gradients are "
                                        "histograms of float atomics)");
  if (p->fwd.batch == 0 || p->total_rows == 0) return HSTU_OK;
  if (attn_bwd_workspace_bytes(*p) > 0 && !p->workspace)
    return set_error(HSTU_EINVAL, "hstu_attn_bwd: this shape needs %zu bytes of workspace", attn_bwd_workspace_bytes(*p));
  hipStream_t st = (hipStream_t)stream;
  switch (p->fwd.dtype) {
    case HSTU_DTYPE_BF16: return launch_attn_bwd_bf16(*p, st);
    case HSTU_DTYPE_F16: return launch_attn_bwd_f16(*p, st);
    default: return launch_attn_bwd_f32(*p, st);
  }
}

}  // extern "C"
