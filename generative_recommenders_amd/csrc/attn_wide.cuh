// Launcher of the wide backward schedule (hstu_attn_bwd_wide.cuh): head dim 128, 16-bit I/O, four 512-register waves.
#pragma once
#include "capi_internal.h"
#include "hstu_attn_bwd_wide.cuh"

namespace hstu {

template <typename T>
static int launch_bwd_wide_dtype(const HstuAttnBwdParams& bp, hipStream_t st) {
  using W = WideCfg<T, 128>;
  const HstuAttnParams& p = bp.fwd;
  if (p.dqk != 128 || p.dv != 128) return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd(wide): head dim %d not instantiated", p.dqk);
  const int tmax = (p.max_seq_len + 31) / 32;
  const int smem = W::smem_bytes();
  static_assert(W::smem_bytes() <= kLdsBudget, "one workgroup per CU");
  auto kern = hstu_attn_bwd_wide_kernel<T, 128>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  int grid = p.batch * p.heads;
  if (WIDE_PERSIST) {
    static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
    if (grid > n_cu) grid = n_cu;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kWideThreads), smem, st, bp, tmax);
  return check_launch("hstu_attn_bwd(wide)");
}

}  // namespace hstu
