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

// HSTU_SOLO_SPLIT=0: the backward as one launch for every length (A/B)
static bool solo_split_enabled() {
  static const bool split = [] { const char* e = getenv("HSTU_SOLO_SPLIT"); return !(e && e[0] == '0'); }();
  return split;
}
// (the forward kernels stay ONE launch: splitting them by length class like the backward ones measured -1 % (plain) and +14 % (bias: 168
// registers for three waves per SIMD cost 19 spills) on the Amazon-Books batch -- a forward wave's time is instruction issue, not latency)
template <typename T>
static int launch_fwd_solo(const HstuAttnParams& p, hipStream_t st) {
  const int smem = kSoloWaves * SoloCfg<T>::fwd_slice(2);
  hipLaunchKernelGGL((hstu_attn_fwd_solo_kernel<T, 2>), dim3(solo_grid(p.batch * p.heads, 3)), dim3(kSoloThreads), smem, st, p, 0, kSoloMaxLen);
  return check_launch("hstu_attn_fwd(solo)");
}

// two launches by length class (hstu_attn_solo.cuh): problems of 33 .. 64 rows with the full slices (two workgroups per CU), problems of
// <= 32 rows with one tile per tensor (three per CU); HSTU_SOLO_SPLIT=0: one launch for everything (A/B)
template <typename T>
static int launch_bwd_solo(const HstuAttnBwdParams& bp, hipStream_t st) {
  const bool split = solo_split_enabled();
  const int total = bp.fwd.batch * bp.fwd.heads;
  const int smem2 = kSoloWaves * SoloCfg<T>::bwd_slice(2), smem1 = kSoloWaves * SoloCfg<T>::bwd_slice(1);
  hipError_t e = hipFuncSetAttribute((const void*)hstu_attn_bwd_solo_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: cannot reserve %d bytes of LDS: %s", smem2, hipGetErrorString(e));
  if (!split) {
    hipLaunchKernelGGL((hstu_attn_bwd_solo_kernel<T, 2>), dim3(solo_grid(total, 2)), dim3(kSoloThreads), smem2, st, bp, 0, kSoloMaxLen);
    return check_launch("hstu_attn_bwd(solo)");
  }
  // (the two launches one after the other: putting the long class on a side stream of the device -- fork / join by events -- measured
  // SLOWER, 69 -> 83 us on the Amazon-Books batch: profiles/r06_ab_solo_length_classes.txt)
  if (bp.fwd.max_seq_len > 32) {
    hipLaunchKernelGGL((hstu_attn_bwd_solo_kernel<T, 2>), dim3(solo_grid(total, 2)), dim3(kSoloThreads), smem2, st, bp, 32, kSoloMaxLen);
    if (int rc = check_launch("hstu_attn_bwd(solo)")) return rc;
  }
  hipLaunchKernelGGL((hstu_attn_bwd_solo_kernel<T, 1>), dim3(solo_grid(total, SOLO_BWD_SHORT_WAVES)), dim3(kSoloThreads), smem1, st, bp, 0, 32);
  return check_launch("hstu_attn_bwd(solo)");
}

// research path (relative bias) at the short-sequence shapes.  Forward: a wave per user (its own tables and bucket bytes).
// HSTU_SOLO_BIAS_SPLIT=0: the backward as one launch for every length (A/B)
static bool solo_bias_split_enabled() {
  static const bool split = [] { const char* e = getenv("HSTU_SOLO_BIAS_SPLIT"); return !(e && e[0] == '0'); }();
  return split;
}
template <typename T, int TPT>
static int launch_fwd_solo_bias_class(const HstuAttnParams& p, hipStream_t st, int len_lo, int len_hi, int max_per_cu) {
  const int tables = bias_table_bytes(p.max_seq_len, p.num_buckets);
  const int smem = kSoloWaves * (SoloCfg<T>::fwd_slice(TPT) + tables + (TPT == 1 ? 1024 : kSoloBucketBytes));
  const int n_cu = cu_count();
  auto kern = hstu_attn_fwd_solo_bias_kernel<T, TPT>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_fwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  }
  const int per_cu = kLdsBudget / smem < max_per_cu ? kLdsBudget / smem : max_per_cu;
  const int wgs = (p.batch + kSoloWaves - 1) / kSoloWaves;
  const int grid = wgs < per_cu * n_cu ? wgs : per_cu * n_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kSoloThreads), smem, st, p, tables, len_lo, len_hi);
  return check_launch("hstu_attn_fwd(solo, bias)");
}
template <typename T>
static int launch_fwd_solo_bias(const HstuAttnParams& p, hipStream_t st) {
  return launch_fwd_solo_bias_class<T, 2>(p, st, 0, kSoloMaxLen, 2);
}

// backward: a workgroup per user at a time, its waves the heads; two launches by length class (hstu_attn_solo.cuh): users of
// 33 .. 64 rows with the full slices (one workgroup per CU), users of <= 32 rows with one tile per tensor (two per CU)
template <typename T>
static int launch_bwd_solo_bias(const HstuAttnBwdParams& bp, hipStream_t st) {
  const HstuAttnParams& p = bp.fwd;
  const int tables = bias_table_bytes(p.max_seq_len, p.num_buckets);
  const int n_cu = cu_count();
  const int hw = 2 * p.max_seq_len + p.num_buckets;
  float* partial = (float*)bp.workspace;
  const bool split = solo_bias_split_enabled();   // 0: one launch, as round 5 (A/B)
  struct Launch { int tpt, len_lo, len_hi, per_cu, ts_copies, hist, smem, grid; } ls[2];
  int nl = 0;
  if (split) {
    if (p.max_seq_len > 32) ls[nl++] = {2, 32, kSoloMaxLen, 1, 1, 0, 0, 0};
    ls[nl++] = {1, 0, 32, 2, 1, 0, 0, 0};
  } else {
    ls[nl++] = {2, 0, kSoloMaxLen, 1, 1, 0, 0, 0};
  }
  int rows = 0, smem_max = 0;
  for (int i = 0; i < nl; ++i) {
    Launch& l = ls[i];
    if (!attn_solo_bias_lds(p, kSoloWaves * SoloCfg<T>::bwd_slice(l.tpt), 2 * (l.tpt == 1 ? 1024 : kSoloBucketBytes), &l.ts_copies, &l.hist, &l.smem))
      return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd(solo, bias): LDS");
    if (l.per_cu * l.smem > kLdsBudget) l.per_cu = 1;
    l.grid = p.batch < l.per_cu * n_cu ? p.batch : l.per_cu * n_cu;
    rows += l.grid;
    if (l.smem > smem_max) smem_max = l.smem;
  }
  hipError_t e = hipSuccess;
  for (const void* k : {(const void*)hstu_attn_bwd_solo_bias_kernel<T, 1>, (const void*)hstu_attn_bwd_solo_bias_kernel<T, 2>}) {
    e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, smem_max);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: cannot reserve %d bytes of LDS: %s", smem_max, hipGetErrorString(e));
  }
  e = hipMemsetAsync(partial, 0, (size_t)rows * hw * sizeof(float), st);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: workspace memset failed: %s", hipGetErrorString(e));
  int row0 = 0;
  for (int i = 0; i < nl; ++i) {
    const Launch& l = ls[i];
    hipStream_t s_i = st;
    if (l.tpt == 1)
      hipLaunchKernelGGL((hstu_attn_bwd_solo_bias_kernel<T, 1>), dim3(l.grid), dim3(kSoloThreads), l.smem, s_i, bp, partial, l.ts_copies, l.hist, tables, l.len_lo, l.len_hi, row0);
    else
      hipLaunchKernelGGL((hstu_attn_bwd_solo_bias_kernel<T, 2>), dim3(l.grid), dim3(kSoloThreads), l.smem, s_i, bp, partial, l.ts_copies, l.hist, tables, l.len_lo, l.len_hi, row0);
    if (int rc = check_launch("hstu_attn_bwd(solo, bias)")) return rc;
    row0 += l.grid;
  }
  return launch_bias_grad_reduce(partial, rows, hw, 2 * p.max_seq_len - 1, bp.dpos_w, bp.dts_w, st);
}

}  // namespace hstu
