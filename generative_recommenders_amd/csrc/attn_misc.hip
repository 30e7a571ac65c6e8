// dtype-independent helpers of the attention ABI.
#include <stdlib.h>

#include "capi_internal.h"
namespace hstu {
int attn_bwd_tiles_bf16(int, int, int, int);
int attn_bwd_tiles_f16(int, int, int, int);
int attn_bwd_tiles_f32(int, int, int, int);

int attn_bwd_tiles_per_block(int dtype, int dqk, int dv, int max_seq_len, int extra_lds) {
  switch (dtype) {
    case HSTU_DTYPE_BF16: return attn_bwd_tiles_bf16(dqk, dv, max_seq_len, extra_lds);
    case HSTU_DTYPE_F16: return attn_bwd_tiles_f16(dqk, dv, max_seq_len, extra_lds);
    default: return attn_bwd_tiles_f32(dqk, dv, max_seq_len, extra_lds);
  }
}

// The folded schedule needs: 16-bit I/O, no bias, no contextual rows, one key block of <= 7 tiles, and head
// dims equal to an instantiated size (its LDS-DMA staging cannot zero-pad).  HSTU_BWD_FOLD=0 forces the
// general kernel (A/B measurements).
bool attn_bwd_fold_applicable(const HstuAttnBwdParams& bp) {
  const HstuAttnParams& p = bp.fwd;
  static const bool enabled = [] { const char* e = getenv("HSTU_BWD_FOLD"); return !(e && e[0] == '0'); }();
  if (!enabled) return false;
  if (p.dtype == HSTU_DTYPE_F32 || p.pos_w || p.contextual_seq_len > 0) return false;
  if (p.dqk != p.dv || (p.dqk != 128 && p.dqk != 64)) return false;
  return (p.max_seq_len + 31) / 32 <= 7;
}

size_t attn_bwd_workspace_bytes(const HstuAttnBwdParams& bp) {
  const HstuAttnParams& p = bp.fwd;
  const int hist = p.pos_w ? ((2 * p.max_seq_len + p.num_buckets) * 4 + 15) / 16 * 16 : 0;
  const int nw = attn_bwd_tiles_per_block(p.dtype, p.dqk, p.dv, p.max_seq_len, hist);
  if (nw <= 0) return 0;
  const int nkb = (p.max_seq_len + 32 * nw - 1) / (32 * nw);
  size_t bytes = 0;
  if (nkb > 1) bytes += ((size_t)bp.total_rows * p.heads * p.dqk * sizeof(float) + 255) / 256 * 256;
  if (p.pos_w) {
    const size_t nblocks = (size_t)((p.batch * p.heads + 7) / 8) * 8 * nkb;
    bytes += nblocks * (2 * p.max_seq_len + p.num_buckets) * sizeof(float);
  }
  return bytes;
}

// column sums of the (rows, width) partial matrix; one thread per column, rows walked in order
// (deterministic across launches)
__global__ void bias_grad_reduce_kernel(const float* partial, int rows, int width, int npos, float* dpos_w, float* dts_w) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= width) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = 0;
  for (; r + 3 < rows; r += 4) {
    s0 += partial[(int64_t)r * width + c];
    s1 += partial[(int64_t)(r + 1) * width + c];
    s2 += partial[(int64_t)(r + 2) * width + c];
    s3 += partial[(int64_t)(r + 3) * width + c];
  }
  for (; r < rows; ++r) s0 += partial[(int64_t)r * width + c];
  const float s = (s0 + s1) + (s2 + s3);
  if (c < npos) dpos_w[c] = s;
  else if (dts_w) dts_w[c - npos] = s;
}

int launch_bias_grad_reduce(const float* partial, int rows, int width, int npos, float* dpos_w, float* dts_w,
                            hipStream_t st) {
  hipLaunchKernelGGL(bias_grad_reduce_kernel, dim3((width + 63) / 64), dim3(64), 0, st, partial, rows, width, npos, dpos_w, dts_w);
  return check_launch("hstu_attn_bwd(bias gradient reduce)");
}
}  // namespace hstu
