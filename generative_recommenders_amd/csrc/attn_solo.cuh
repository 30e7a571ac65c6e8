// Launchers of the short-sequence kernels (hstu_attn_solo.cuh): one wave per (user, head).
#pragma once
#include "capi_internal.h"
#include "hstu_attn_solo.cuh"

namespace hstu {

static int solo_grid(int total, int per_cu) {
  static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
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

}  // namespace hstu
