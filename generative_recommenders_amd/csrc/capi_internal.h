// Internal helpers shared by the translation units of libhstu_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hstu_hip.h"

namespace hstu {
int set_error(int code, const char* fmt, ...);   // records the message, returns `code`
int check_launch(const char* what);              // hipGetLastError() -> HSTU_OK | HSTU_ELAUNCH

// per-dtype attention launchers (one translation unit each, so they compile in parallel)
int launch_attn_fwd_bf16(const HstuAttnParams& p, hipStream_t st);
int launch_attn_fwd_f16(const HstuAttnParams& p, hipStream_t st);
int launch_attn_fwd_f32(const HstuAttnParams& p, hipStream_t st);
int launch_attn_bwd_bf16(const HstuAttnBwdParams& p, hipStream_t st);
int launch_attn_bwd_f16(const HstuAttnBwdParams& p, hipStream_t st);
int launch_attn_bwd_f32(const HstuAttnBwdParams& p, hipStream_t st);
int launch_attn_fwd_bias_bf16(const HstuAttnParams& p, hipStream_t st);
int launch_attn_fwd_bias_f16(const HstuAttnParams& p, hipStream_t st);
int launch_attn_fwd_bias_f32(const HstuAttnParams& p, hipStream_t st);
int launch_attn_bwd_bias_bf16(const HstuAttnBwdParams& p, hipStream_t st);
int launch_attn_bwd_bias_f16(const HstuAttnBwdParams& p, hipStream_t st);
int launch_attn_bwd_bias_f32(const HstuAttnBwdParams& p, hipStream_t st);
// folded schedule of the backward for short sequences (hstu_attn_bwd_fold.cuh); 16-bit dtypes only
int launch_attn_bwd_fold_bf16(const HstuAttnBwdParams& p, hipStream_t st);
int launch_attn_bwd_fold_f16(const HstuAttnBwdParams& p, hipStream_t st);
bool attn_bwd_fold_applicable(const HstuAttnBwdParams& p);
// long sequences (more than one key block), 16-bit I/O, no bias, no contextual rows: dK / dV kernel + dQ kernel, no atomics, no
// workspace (hstu_attn_bwd_long.cuh; HSTU_BWD_LONG=0: the general kernel, A/B measurements)
int launch_attn_bwd_long_bf16(const HstuAttnBwdParams& p, hipStream_t st);
int launch_attn_bwd_long_f16(const HstuAttnBwdParams& p, hipStream_t st);
bool attn_bwd_long_applicable(const HstuAttnBwdParams& p);
// short sequences (max_seq_len <= 64, head dims <= 32, 16-bit I/O): one wave per (user, head) (hstu_attn_solo.cuh; HSTU_SOLO=0 disables)
int launch_attn_fwd_solo_bf16(const HstuAttnParams& p, hipStream_t st);
int launch_attn_fwd_solo_f16(const HstuAttnParams& p, hipStream_t st);
int launch_attn_bwd_solo_bf16(const HstuAttnBwdParams& p, hipStream_t st);
int launch_attn_bwd_solo_f16(const HstuAttnBwdParams& p, hipStream_t st);
bool attn_solo_applicable(const HstuAttnParams& p, bool backward);
// the same shapes with the research path's relative bias (hstu_attn_{fwd,bwd}_solo_bias_kernel; HSTU_SOLO_BIAS=0 disables)
int launch_attn_fwd_solo_bias_bf16(const HstuAttnParams& p, hipStream_t st);
int launch_attn_fwd_solo_bias_f16(const HstuAttnParams& p, hipStream_t st);
int launch_attn_bwd_solo_bias_bf16(const HstuAttnBwdParams& p, hipStream_t st);
int launch_attn_bwd_solo_bias_f16(const HstuAttnBwdParams& p, hipStream_t st);
bool attn_solo_bias_applicable(const HstuAttnParams& p, bool backward);
bool attn_solo_bias_lds(const HstuAttnParams& p, int base, int cache, int* ts_copies, int* hist_bytes, int* smem);
// ... and, among those, the ones the 4-wave / two-workgroups-per-CU kernel takes (head dim 64; HSTU_BWD_QUAD=0 disables)
bool attn_bwd_quad_applicable(const HstuAttnBwdParams& p);
// sums the per-workgroup bias-gradient rows: partial (rows, width) -> dpos_w (npos), dts_w (width - npos)
int launch_bias_grad_reduce(const float* partial, int rows, int width, int npos, float* dpos_w, float* dts_w,
                            hipStream_t st);
size_t attn_bwd_workspace_bytes(const HstuAttnBwdParams& p);
// profiler names of the instantiations the launchers above pick (attn_misc.hip)
int attn_kernel_name(const HstuAttnParams& p, const HstuAttnBwdParams* bwd, char* buf, size_t len);
int attn_bwd_tiles_per_block(int dtype, int dqk, int dv, int max_seq_len, int extra_lds);
// LDS bytes of the research-path bias state of a backward workgroup (histograms with *ts_copies privatised copies of
// the time-bucket histogram + the staged tables) and the number of copies chosen; 0 without bias
bool attn_bias_head_loop_enabled();
// HSTU_ATTN_PRECISE=1: the forward's P' as two 16-bit fragments (value + rounding remainder) for 16-bit I/O at 64 x 64 / 128 x 128
// without bias: output error at the rounding floor of the I/O dtype, ~20-25 % slower (hstu_attn_fwd.cuh, PRECISE)
bool attn_fwd_precise_enabled();
// research-path forward, short sequences: does one workgroup per (user, query block) walk the heads (hstu_attn_fwd.cuh,
// HEADS instantiation)?  ONE decision for the launcher and for attn_kernel_name.  `ring_bytes` = the K/V ring of the
// instantiation (FwdCfg::SMEM); *tables / *cache = LDS bytes of the staged tables and of the bucket bytes.
bool attn_fwd_head_loop_applicable(const HstuAttnParams& p, int ring_bytes, int* tables, int* cache);
int attn_fwd_ring_bytes(int elem_bytes, int dqk_padded, int dv_padded);   // == FwdCfg<T, DQK, DV>::SMEM (checked by the launcher)
bool attn_bwd_fold_bias_lds(const HstuAttnParams& p, int base, int* ts_copies, int* hist_bytes, int* smem);
bool attn_bwd_fold_bias_applicable(const HstuAttnBwdParams& bp);
int attn_bwd_bias_lds(const HstuAttnParams& p, int* ts_copies);

// compute units of the CURRENT device (cached per device: a host may hold devices of different sizes)
inline int cu_count() {
  static int cache[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (cache[dev] == 0) {
    int n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cache[dev] = n > 0 ? n : 256;
  }
  return cache[dev];
}
inline int pad_head_dim(int d) { return d <= 32 ? 32 : (d <= 64 ? 64 : (d <= 128 ? 128 : 0)); }
constexpr int kLdsBudget = 160 * 1024;
// bytes of the bias tables a workgroup stages in LDS: pos_w (2N-1 floats), ts_w (nb+1 floats), N int64 timestamps,
// their int32 offsets
inline int bias_table_bytes(int max_seq_len, int num_buckets) {
  return 128 /* zeros in front of the position table: see stage_bias_tables */ +
         ((2 * max_seq_len * 4 + 15) / 16 + ((num_buckets + 1) * 4 + 15) / 16 + (max_seq_len * 8 + 15) / 16) * 16 +
         (2 * ((max_seq_len + 32 + 3) / 4 * 4) * 4 + 8 * 4 + 15) / 16 * 16;   // + int32 offsets and their copy shifted by one (padded), one range flag per wave
}
constexpr int kDqScratchBytes = 8 * 4096;   // general backward, several key blocks: one [32 q][32 d] fp32 tile per wave
}  // namespace hstu
