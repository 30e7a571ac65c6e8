// Launcher of the folded backward schedule (hstu_attn_bwd_fold.cuh).
#pragma once
// The folded and four-wave backward kernels read every tile of q / k / v / dO exactly once, from one CU: their LDS-DMA requests
// carry the non-temporal hint (-0.5 % M-full, -1.4 % M-jag, bit-identical: profiles/r04_dma_nt.txt).  The forward must NOT: its
// query blocks share a problem's K / V through L2 (+18..24 % with the hint).
#ifndef HSTU_DMA_NT
#define HSTU_DMA_NT 1
#endif
#include "capi_internal.h"
#include "hstu_attn_bwd_quad.cuh"

namespace hstu {

template <typename T, int D>
static int launch_bwd_fold_inst(const HstuAttnBwdParams& bp, hipStream_t st) {
  using F = FoldCfg<T, D, D>;
  const HstuAttnParams& p = bp.fwd;
  const int tmax = (p.max_seq_len + 31) / 32;
  const int smem = F::smem_bytes();
  auto kern = hstu_attn_bwd_fold_kernel<T, D, D>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  }
  // one persistent workgroup per CU (never more workgroups than problems: every workgroup reads its first problem's offsets)
  int grid = p.batch * p.heads;
  const int n_cu = cu_count();
  if (grid > n_cu) grid = n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBwdThreads), smem, st, bp, tmax);
  return check_launch("hstu_attn_bwd(fold)");
}

// head dim 64: 4-wave workgroups, two per CU (hstu_attn_bwd_quad.cuh)
template <typename T, int D>
static int launch_bwd_quad_inst(const HstuAttnBwdParams& bp, hipStream_t st) {
  using Q = QuadCfg<T, D>;
  const HstuAttnParams& p = bp.fwd;
  const int tmax = (p.max_seq_len + 31) / 32;
  const int smem = Q::smem_bytes();
  static_assert(Q::smem_bytes() <= 80 * 1024, "two workgroups per CU");
  auto kern = hstu_attn_bwd_quad_kernel<T, D>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  int grid = p.batch * p.heads;
  if (QUAD_PERSIST) {
    const int n_cu = cu_count();
    if (grid > 2 * n_cu) grid = 2 * n_cu;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kQuadThreads), smem, st, bp, tmax);
  return check_launch("hstu_attn_bwd(quad)");
}

template <typename T>
static int launch_bwd_fold_dtype(const HstuAttnBwdParams& bp, hipStream_t st) {
  if (bp.fwd.dqk == 128) return launch_bwd_fold_inst<T, 128>(bp, st);
  if (bp.fwd.dqk == 64 && attn_bwd_quad_applicable(bp)) return launch_bwd_quad_inst<T, 64>(bp, st);
  if (bp.fwd.dqk == 64) return launch_bwd_fold_inst<T, 64>(bp, st);
  return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd(fold): head dim %d not instantiated", bp.fwd.dqk);
}

}  // namespace hstu
