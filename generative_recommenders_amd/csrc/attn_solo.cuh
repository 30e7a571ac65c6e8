// Launchers of the short-sequence kernels (hstu_attn_solo.cuh): one wave per (user, head).
#pragma once
#include "capi_internal.h"
#include <stdlib.h>

#include "hstu_attn_solo.cuh"

namespace hstu {

static int solo_grid(int total, int per_cu) {
  const int n_cu = cu_count();
  const int wgs = (total + kSoloWaves - 1) / kSoloWaves;
  return wgs < per_cu * n_cu ? wgs : per_cu * n_cu;      // as many workgroups as fit a CU (LDS) walk the problems
}

template <typename T>
static int launch_fwd_solo(const HstuAttnParams& p, hipStream_t st) {
  const int smem = kSoloWaves * SoloCfg<T>::fwd_slice();
  hipLaunchKernelGGL(hstu_attn_fwd_solo_kernel<T>, dim3(solo_grid(p.batch * p.heads, 3)), dim3(kSoloThreads), smem, st, p);
  return check_launch("hstu_attn_fwd(solo)");
}

template <typename T>
static int launch_bwd_solo(const HstuAttnBwdParams& bp, hipStream_t st) {
  const int smem = kSoloWaves * SoloCfg<T>::bwd_slice();
  auto kern = hstu_attn_bwd_solo_kernel<T>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  hipLaunchKernelGGL(kern, dim3(solo_grid(bp.fwd.batch * bp.fwd.heads, 2)), dim3(kSoloThreads), smem, st, bp);
  return check_launch("hstu_attn_bwd(solo)");
}

// research path (relative bias) at the short-sequence shapes.  Forward: a wave per user (its own tables and bucket bytes)
template <typename T>
static int launch_fwd_solo_bias(const HstuAttnParams& p, hipStream_t st) {
  const int tables = bias_table_bytes(p.max_seq_len, p.num_buckets);
  const int smem = kSoloWaves * (SoloCfg<T>::fwd_slice() + tables + kSoloBucketBytes);
  const int n_cu = cu_count();
  auto kern = hstu_attn_fwd_solo_bias_kernel<T>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_fwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  }
  const int per_cu = kLdsBudget / smem < 3 ? kLdsBudget / smem : 3;
  const int wgs = (p.batch + kSoloWaves - 1) / kSoloWaves;
  const int grid = wgs < per_cu * n_cu ? wgs : per_cu * n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kSoloThreads), smem, st, p, tables);
  return check_launch("hstu_attn_fwd(solo, bias)");
}

// backward: a workgroup per user at a time, its waves the heads
template <typename T>
static int launch_bwd_solo_bias(const HstuAttnBwdParams& bp, hipStream_t st) {
  const HstuAttnParams& p = bp.fwd;
  int ts_copies = 1, hist = 0, smem = 0;
  if (!attn_solo_bias_lds(p, kSoloWaves * SoloCfg<T>::bwd_slice(), 2 * kSoloBucketBytes, &ts_copies, &hist, &smem))
    return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd(solo, bias): LDS");
  const int tables = bias_table_bytes(p.max_seq_len, p.num_buckets);
  auto kern = hstu_attn_bwd_solo_bias_kernel<T>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  const int n_cu = cu_count();
  const int grid = p.batch < n_cu ? p.batch : n_cu;
  const int hw = 2 * p.max_seq_len + p.num_buckets;
  float* partial = (float*)bp.workspace;
  e = hipMemsetAsync(partial, 0, (size_t)grid * hw * sizeof(float), st);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: workspace memset failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kSoloThreads), smem, st, bp, partial, ts_copies, hist, tables);
  if (int rc = check_launch("hstu_attn_bwd(solo, bias)")) return rc;
  return launch_bias_grad_reduce(partial, grid, hw, 2 * p.max_seq_len - 1, bp.dpos_w, bp.dts_w, st);
}

}  // namespace hstu
