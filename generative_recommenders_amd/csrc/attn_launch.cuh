// Per-dtype launchers: pick the (padded) head-dim instantiation and launch.
#pragma once
#include "capi_internal.h"
#include "hstu_attn_bwd_fold.cuh"

namespace hstu {

template <typename T, int DQK, int DV, bool BIAS = false>
static int launch_fwd_inst(const HstuAttnParams& p, hipStream_t st) {
  using C = FwdCfg<T, DQK, DV>;
  const int q_rows = p.delta_q > 0 ? p.delta_q : p.max_seq_len;
  const int nqb = (q_rows + kFwdRowsPerBlock - 1) / kFwdRowsPerBlock;
  auto kern = hstu_attn_fwd_kernel<T, DQK, DV, BIAS, false>;
  // research-path bias, short sequences: one workgroup per (user, query block) walks the heads with the time buckets of
  // its tile pairs as bytes in LDS (hstu_attn_fwd.cuh); the decision is attn_misc.hip's (shared with attn_kernel_name)
  int tables = 0, cache = 0;
  if (attn_fwd_ring_bytes((int)sizeof(T), DQK, DV) != C::SMEM)
    return set_error(HSTU_ELAUNCH, "hstu_attn_fwd: attn_fwd_ring_bytes is out of step with FwdCfg (%d vs %d)", attn_fwd_ring_bytes((int)sizeof(T), DQK, DV), C::SMEM);
  const bool hl_ok = attn_fwd_head_loop_applicable(p, C::SMEM, &tables, &cache);   // (also fills tables / cache)
  const bool head_loop = BIAS && sizeof(T) == 2 && hl_ok;
  if (!BIAS) tables = 0;
  const int groups = ((head_loop ? p.batch : p.batch * p.heads) + 7) / 8;
  if constexpr (BIAS && sizeof(T) == 2) {   // (fp32 I/O: the plain kernel; its head-loop variant spills at 128 x 128)
    if (head_loop) kern = hstu_attn_fwd_kernel<T, DQK, DV, true, true>;
  }
  if constexpr (!BIAS && sizeof(T) == 2 && DQK == DV && (DQK == 64 || DQK == 128)) {
    if (attn_fwd_precise_enabled()) kern = hstu_attn_fwd_kernel<T, DQK, DV, false, false, true>;
  }
  const int smem = C::SMEM + tables + (head_loop ? cache : 0);
  if (smem > kLdsBudget) return set_error(HSTU_EUNSUPPORTED, "hstu_attn_fwd: max_seq_len %d needs %d bytes of LDS for the bias tables", p.max_seq_len, smem);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_fwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3(groups * 8 * nqb), dim3(kFwdThreads), smem, st, p, nqb, head_loop ? tables : 0);
  return check_launch("hstu_attn_fwd");
}

template <typename T, int DQK, int DV, bool BIAS>
static int launch_bwd_inst(const HstuAttnBwdParams& bp, hipStream_t st);

template <typename T>
static int launch_fwd_bias_dtype(const HstuAttnParams& p, hipStream_t st) {
  const int a = pad_head_dim(p.dqk), v = pad_head_dim(p.dv);
#define CASE(A) if (a == A && v == A) return launch_fwd_inst<T, A, A, true>(p, st);
  CASE(32) CASE(64) CASE(128)
#undef CASE
  return set_error(HSTU_EUNSUPPORTED, "hstu_attn_fwd: relative-bias attention is instantiated for dqk == dv in {32, 64, 128} (got %d, %d)", p.dqk, p.dv);
}

template <typename T>
static int launch_bwd_bias_dtype(const HstuAttnBwdParams& bp, hipStream_t st) {
  const int a = pad_head_dim(bp.fwd.dqk), v = pad_head_dim(bp.fwd.dv);
#define CASE(A) if (a == A && v == A) return launch_bwd_inst<T, A, A, true>(bp, st);
  CASE(32) CASE(64) CASE(128)
#undef CASE
  return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd: relative-bias attention is instantiated for dqk == dv in {32, 64, 128} (got %d, %d)", bp.fwd.dqk, bp.fwd.dv);
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
static int bwd_tiles_inst(int max_seq_len, int extra_lds) {
  using C = BwdCfg<T, DQK, DV>;
  int nw = C::max_tiles(kLdsBudget - extra_lds);
  const int need = (max_seq_len + 31) / 32;
  if (need <= nw) return need < 1 ? 1 : need;        // one key block: no dq accumulation, no scratch
  // several key blocks: the helpers' fp32 dq partials go through a per-wave LDS scratch tile (kDqScratchBytes)
  nw = C::max_tiles(kLdsBudget - extra_lds - kDqScratchBytes);
  // Away from the diagonal every key tile of the block is active in every step: nw owner waves run one pair each while the
  // 8 - nw others run the dQ GEMM of the previous query tile, dealt in whole 32-feature blocks (at most DQK/32 helpers
  // have work).  With the block as large as the LDS allows the owners wait for them most of every step (cycle trace,
  // N = 1024, 128-wide heads, 5 tiles: owners 6 K cycles, helpers 12.5 K), and a sequence that does not divide into
  // such blocks ends in a block of one or two tiles whose workgroup is mostly idle waves.  Measured (tools/long_bwd.py
  // with HSTU_BWD_NW = 3..7 at N = 256 .. 8192): at most 6 tiles per block at head dim 64 (N = 8192: 13.0 -> 10.2 ms),
  // 5 at 32 (N = 1024: 5.7 -> 4.0 ms), and the tiles spread EVENLY over the fewest blocks that takes (16 tiles at head
  // dim 128 = 4 x 4 instead of 5 + 5 + 5 + 1: 11.8 -> 9.0 ms).  Dealing single (feature block, key tile) contractions
  // to the helpers instead -- every one busy -- was measured too and is slower: each partial sum pays the LDS transpose
  // and 16 atomic instructions of its own.
  static const int forced = [] { const char* e = getenv("HSTU_BWD_NW"); return e ? atoi(e) : 0; }();
  const int cap = DQK >= 128 ? 5 : (DQK >= 64 ? 6 : 5);
  if (nw > cap) nw = cap;
  if (nw < 1) nw = 1;
  const int blocks = (need + nw - 1) / nw;
  nw = (need + blocks - 1) / blocks;
  if (forced > 0 && forced < nw) nw = forced;
  return nw < 1 ? 1 : nw;
}

// research-path bias on the folded schedule (hstu_attn_bwd_fold_bias_kernel): head dim 64, 16-bit I/O
template <typename T, int D>
static int launch_bwd_fold_bias_inst(const HstuAttnBwdParams& bp, hipStream_t st) {
  using F = FoldCfg<T, D, D>;
  const HstuAttnParams& p = bp.fwd;
  const int tmax = (p.max_seq_len + 31) / 32;
  int ts_copies = 1, hist = 0, smem = 0;
  if (!attn_bwd_fold_bias_lds(p, F::smem_bytes(), &ts_copies, &hist, &smem)) return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd(fold, bias): LDS");
  const int tables = bias_table_bytes(p.max_seq_len, p.num_buckets);
  auto kern = hstu_attn_bwd_fold_bias_kernel<T, D>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  const int n_cu = cu_count();
  const int grid = p.batch < n_cu ? p.batch : n_cu;
  const int hw = 2 * p.max_seq_len + p.num_buckets;
  float* partial = (float*)bp.workspace;
  // (the user counter of the dynamic hand-out sits right behind the partial rows, in the first word of the region the reduce's chunk sums
  // take over AFTER this kernel: one memset zeroes both)
  int* const next_user = (int*)(partial + (size_t)grid * hw);
  e = hipMemsetAsync(partial, 0, ((size_t)grid * hw + 1) * sizeof(float), st);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: workspace memset failed: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBwdThreads), smem, st, bp, tmax, partial, ts_copies, hist, tables, next_user);
  if (int rc = check_launch("hstu_attn_bwd(fold, bias)")) return rc;
  return launch_bias_grad_reduce(partial, grid, hw, 2 * p.max_seq_len - 1, bp.dpos_w, bp.dts_w, st);
}

template <typename T, int DQK, int DV, bool BIAS>
static int launch_bwd_inst(const HstuAttnBwdParams& bp, hipStream_t st) {
  using C = BwdCfg<T, DQK, DV>;
  const HstuAttnParams& p = bp.fwd;
  if constexpr (BIAS && sizeof(T) == 2 && DQK == 64 && DV == 64) {
    if (attn_bwd_fold_bias_applicable(bp)) return launch_bwd_fold_bias_inst<T, 64>(bp, st);
  }
  int ts_copies = 1;
  const int hist = attn_bwd_bias_lds(p, &ts_copies);
  const int nw = bwd_tiles_inst<T, DQK, DV>(p.max_seq_len, hist);
  const int nkb = (p.max_seq_len + 32 * nw - 1) / (32 * nw);
  // research-path bias with the whole sequence in one key block: one workgroup per USER walks the heads and keeps the
  // time-bucket matrix as bytes in LDS (1 KiB per tile pair of the causal triangle), see hstu_attn_bwd.cuh
  const int cache_bytes = nw * (nw + 1) / 2 * 1024;
  const bool head_loop = BIAS && nkb == 1 && p.heads > 1 && p.ts_w && p.timestamps && p.num_buckets <= 255 && p.contextual_seq_len == 0 &&
                         C::smem_bytes(nw, hist + cache_bytes) <= kLdsBudget && attn_bias_head_loop_enabled();
  const int groups = ((head_loop ? p.batch : p.batch * p.heads) + 7) / 8;
  const int nblocks = groups * 8 * nkb;
  const int smem = C::smem_bytes(nw, hist + (nkb > 1 ? kDqScratchBytes : 0) + (head_loop ? cache_bytes : 0));
  if (smem > kLdsBudget) return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd: max_seq_len %d needs %d bytes of LDS for the bias histograms", p.max_seq_len, smem);
  auto kern = hstu_attn_bwd_kernel<T, DQK, DV, BIAS>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  }
  // workspace layout: [fp32 dq accumulator (several key blocks only)] [bias-gradient partial rows]
  float* acc = nullptr;
  size_t acc_bytes = 0;
  if (BIAS && bp.deterministic)
    return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd: deterministic = 1 is not available with the relative bias (the table gradients are "
                                        "histograms of float atomics)");
  const size_t slab_bytes = ((size_t)bp.total_rows * p.heads * p.dqk * sizeof(float) + 255) / 256 * 256;
  const int n_slabs = (nkb > 1 && bp.deterministic) ? nkb : 1;    // deterministic: key block kb stores (not adds) into slab kb
  if (nkb > 1) {
    acc = (float*)bp.workspace;
    acc_bytes = slab_bytes * n_slabs;
    hipError_t e = hipMemsetAsync(acc, 0, acc_bytes, st);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: workspace memset failed: %s", hipGetErrorString(e));
  }
  float* partial = nullptr;
  const int hw = 2 * p.max_seq_len + p.num_buckets;
  if (BIAS) {
    partial = (float*)((char*)bp.workspace + acc_bytes);
    hipError_t e = hipMemsetAsync(partial, 0, (size_t)nblocks * hw * sizeof(float), st);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd: workspace memset failed: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(kBwdThreads), smem, st, bp, nkb, nw, acc, partial, ts_copies, head_loop ? hist : 0,
                     n_slabs > 1 ? (int64_t)(slab_bytes / sizeof(float)) : (int64_t)0);
  if (int e = check_launch("hstu_attn_bwd")) return e;
  if (nkb > 1) {
    const int64_t n = bp.total_rows * p.heads * (int64_t)(p.dqk / (16 / Elem<T>::kBytes));   // 16 bytes of dq per thread
    int blocks = (int)((n + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(hstu_dq_convert_kernel<T>, dim3(blocks), dim3(256), 0, st, acc, bp.dq, bp.total_rows, p.heads,
                       p.dqk, bp.dq_row_stride, bp.dq_head_stride, n_slabs, (int64_t)(slab_bytes / sizeof(float)));
    if (int e = check_launch("hstu_attn_bwd(dq convert)")) return e;
  }
  if (BIAS) return launch_bias_grad_reduce(partial, nblocks, hw, 2 * p.max_seq_len - 1, bp.dpos_w, bp.dts_w, st);
  return HSTU_OK;
}

template <typename T>
static int launch_bwd_dtype(const HstuAttnBwdParams& bp, hipStream_t st) {
  const int a = pad_head_dim(bp.fwd.dqk), v = pad_head_dim(bp.fwd.dv);
#define CASE(A, V) if (a == A && v == V) return launch_bwd_inst<T, A, V, false>(bp, st);
  CASE(32, 32) CASE(32, 64) CASE(32, 128) CASE(64, 32) CASE(64, 64) CASE(64, 128) CASE(128, 32) CASE(128, 64) CASE(128, 128)
#undef CASE
  return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd: head dims (%d, %d) not instantiated", bp.fwd.dqk, bp.fwd.dv);
}

template <typename T>
static int bwd_tiles_dtype(int dqk, int dv, int max_seq_len, int extra_lds) {
  const int a = pad_head_dim(dqk), v = pad_head_dim(dv);
#define CASE(A, V) if (a == A && v == V) return bwd_tiles_inst<T, A, V>(max_seq_len, extra_lds);
  CASE(32, 32) CASE(32, 64) CASE(32, 128) CASE(64, 32) CASE(64, 64) CASE(64, 128) CASE(128, 32) CASE(128, 64) CASE(128, 128)
#undef CASE
  return 0;
}

}  // namespace hstu
