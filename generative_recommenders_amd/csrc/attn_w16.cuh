// Launcher of the sixteen-wave backward (hstu_attn_bwd_w16.cuh): head dim 128, 16-bit I/O, one persistent workgroup per CU.
#pragma once
#ifndef HSTU_DMA_NT
#define HSTU_DMA_NT 1      // every tile is read once, from one CU: non-temporal LDS-DMA requests (as attn_fold.cuh)
#endif
#include "capi_internal.h"
#include "hstu_attn_bwd_w16.cuh"

namespace hstu {

template <typename T>
static int launch_bwd_w16_dtype(const HstuAttnBwdParams& bp, hipStream_t st) {
  using F = W16Cfg<T, 128>;
  const HstuAttnParams& p = bp.fwd;
  if (p.dqk != 128 || p.dv != 128) return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd(w16): head dim %d not instantiated", p.dqk);
  const int tmax = (p.max_seq_len + 31) / 32;
  const int smem = F::smem_bytes();
  static_assert(F::smem_bytes() <= kLdsBudget, "one workgroup per CU");
  auto kern = hstu_attn_bwd_w16_kernel<T, 128>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  int grid = p.batch * p.heads;
  if (grid > n_cu) grid = n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kW16Threads), smem, st, bp, tmax);
  return check_launch("hstu_attn_bwd(w16)");
}

}  // namespace hstu
