// Per-dtype launchers: pick the (padded) head-dim instantiation and launch.
#pragma once
#include "capi_internal.h"
#include "hstu_attn_bwd.cuh"

namespace hstu {

template <typename T, int DQK, int DV>
static int launch_fwd_inst(const HstuAttnParams& p, hipStream_t st) {
  using C = FwdCfg<T, DQK, DV>;
  const int q_rows = p.delta_q > 0 ? p.delta_q : p.max_seq_len;
  const int nqb = (q_rows + kFwdRowsPerBlock - 1) / kFwdRowsPerBlock;
  const int groups = (p.batch * p.heads + 7) / 8;
  auto kern = hstu_attn_fwd_kernel<T, DQK, DV>;
  if (C::SMEM > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_fwd: cannot reserve %d bytes of LDS: %s", C::SMEM, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3(groups * 8 * nqb), dim3(kFwdThreads), C::SMEM, st, p, nqb);
  return check_launch("hstu_attn_fwd");
}

template <typename T>
static int launch_fwd_dtype(const HstuAttnParams& p, hipStream_t st) {
  const int a = pad_head_dim(p.dqk), v = pad_head_dim(p.dv);
#define CASE(A, V) if (a == A && v == V) return launch_fwd_inst<T, A, V>(p, st);
  CASE(32, 32) CASE(32, 64) CASE(32, 128) CASE(64, 32) CASE(64, 64) CASE(64, 128) CASE(128, 32) CASE(128, 64) CASE(128, 128)
#undef CASE
  return set_error(HSTU_EUNSUPPORTED, "hstu_attn_fwd: head dims (%d, %d) not instantiated", p.dqk, p.dv);
}

template <typename T, int DQK, int DV>
static int bwd_tiles_inst(int max_seq_len) {
  using C = BwdCfg<T, DQK, DV>;
  int nw = C::max_tiles(kLdsBudget);
  const int need = (max_seq_len + 31) / 32;
  if (need < nw) nw = need;
  return nw < 1 ? 1 : nw;
}

template <typename T, int DQK, int DV>
static int launch_bwd_inst(const HstuAttnBwdParams& bp, hipStream_t st) {
  using C = BwdCfg<T, DQK, DV>;
  const HstuAttnParams& p = bp.fwd;
  const int nw = bwd_tiles_inst<T, DQK, DV>(p.max_seq_len);
  const int nkb = (p.max_seq_len + 32 * nw - 1) / (32 * nw);
  const int groups = (p.batch * p.heads + 7) / 8;
  const int smem = C::smem_bytes(nw);
  auto kern = hstu_attn_bwd_kernel<T, DQK, DV>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  }
  float* acc = nullptr;
  if (nkb > 1) {
    acc = (float*)bp.workspace;
    const size_t bytes = (size_t)bp.total_rows * p.heads * p.dqk * sizeof(float);
    hipError_t e = hipMemsetAsync(acc, 0, bytes, st);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: workspace memset failed: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3(groups * 8 * nkb), dim3(kBwdThreads), smem, st, bp, nkb, nw, acc);
  if (int e = check_launch("hstu_attn_bwd")) return e;
  if (nkb > 1) {
    const int64_t n = bp.total_rows * p.heads * (int64_t)p.dqk;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(hstu_dq_convert_kernel<T>, dim3(blocks), dim3(256), 0, st, acc, bp.dq, bp.total_rows, p.heads,
                       p.dqk, bp.dq_row_stride, bp.dq_head_stride);
    return check_launch("hstu_attn_bwd(dq convert)");
  }
  return HSTU_OK;
}

template <typename T>
static int launch_bwd_dtype(const HstuAttnBwdParams& bp, hipStream_t st) {
  const int a = pad_head_dim(bp.fwd.dqk), v = pad_head_dim(bp.fwd.dv);
#define CASE(A, V) if (a == A && v == V) return launch_bwd_inst<T, A, V>(bp, st);
  CASE(32, 32) CASE(32, 64) CASE(32, 128) CASE(64, 32) CASE(64, 64) CASE(64, 128) CASE(128, 32) CASE(128, 64) CASE(128, 128)
#undef CASE
  return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd: head dims (%d, %d) not instantiated", bp.fwd.dqk, bp.fwd.dv);
}

template <typename T>
static int bwd_tiles_dtype(int dqk, int dv, int max_seq_len) {
  const int a = pad_head_dim(dqk), v = pad_head_dim(dv);
#define CASE(A, V) if (a == A && v == V) return bwd_tiles_inst<T, A, V>(max_seq_len);
  CASE(32, 32) CASE(32, 64) CASE(32, 128) CASE(64, 32) CASE(64, 64) CASE(64, 128) CASE(128, 32) CASE(128, 64) CASE(128, 128)
#undef CASE
  return 0;
}

}  // namespace hstu
