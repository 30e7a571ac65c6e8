// Timestamp / position additive encoder (HBM-bound): the step right before the STU stack.
//   out[row] = alpha * x[row] + pos_w[pos_idx(row)] + ts_w[ts_idx(row)]
// Reference semantics: ops/position.py:38-96, ops/pytorch/pt_position.py:40-134 (index arithmetic restated below,
// bit-exact: integer position index, fp32 time bucket); kernels replaced: ops/triton/triton_position.py:62-158
// (forward), :188-238 (table gradients).  Caller: modules/positional_encoder.py:52-75.
//
// Forward: one workgroup per user (offset / length / query-time loads are scalar), one wave per row, 16 bytes of x
// per lane, fp32 tables (32 bytes per lane each), fp32 math, one rounding to the I/O dtype; the two table indices of
// every row are written out for the backward.
// Backward (table gradients = sums of dout rows per table row): embedding_grad.hip.
#include "hstu_common.cuh"
#include "capi_internal.h"

namespace hstu {

constexpr int kPosThreads = 256;

// position-table row of row r of a user of length len (pt_position.py:40-73)
HSTU_DEV int pos_index(int r, int len, int nt, int has_targets, int interleave, int max_ctx, int max_pos_ind) {
  int idx;
  if (has_targets) {
    const int high = len - nt * (interleave ? 2 : 1);
    idx = high - min(r, high);
  } else {
    idx = len - r;
  }
  idx = min(idx + max_ctx, max_pos_ind - 1);
  if (r < max_ctx) idx = r;
  return idx;
}

// time bucket of (query_time - t) (pt_position.py:100-122): int64 difference -> fp32, clamp(min = 1e-6), / 60,
// sqrt | log, truncate, clamp to [0, max_bucket]
HSTU_DEV int time_bucket(int64_t query_time, int64_t t, int use_log, int max_bucket) {
  float d = (float)(query_time - t);
  d = fmaxf(d, 1e-6f) / 60.0f;
  d = use_log ? logf(d) : sqrtf(d);
  d = fmaxf(d, 0.0f);
  const int bkt = (int)d;
  return min(max(bkt, 0), max_bucket);
}

template <typename T>
__global__ __launch_bounds__(kPosThreads) void add_ts_pos_fwd_kernel(const T* x, T* out, const void* seq_offsets,
                                                                     const int64_t* timestamps, const void* num_targets,
                                                                     const float* pos_w, const float* ts_w, int32_t* pos_idx,
                                                                     int32_t* ts_idx, int dim, int max_ctx, int max_pos_ind,
                                                                     int max_bucket, int interleave, int use_log, float alpha,
                                                                     int is64) {
  constexpr int VEC = 16 / sizeof(T);
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t off = load_index(seq_offsets, b, is64);
  const int len = (int)(load_index(seq_offsets, b + 1, is64) - off);
  if (len <= 0) return;
  const int nt = num_targets ? (int)load_index(num_targets, b, is64) : 0;
  const int64_t query_time = timestamps[off + len - 1];
  for (int r = wave; r < len; r += kPosThreads / 64) {
    const int64_t row = off + r;
    const int pi = pos_index(r, len, nt, num_targets != nullptr, interleave, max_ctx, max_pos_ind);
    const int ti = time_bucket(query_time, timestamps[row], use_log, max_bucket);
    if (lane == 0) { pos_idx[row] = pi; ts_idx[row] = ti; }
    const T* xr = x + row * dim;
    T* orow = out + row * dim;
    const float* pr = pos_w + (int64_t)pi * dim;
    const float* tr = ts_w + (int64_t)ti * dim;
    for (int c = lane * VEC; c < dim; c += 64 * VEC) {
      if (c + VEC <= dim) {
        typedef T tv __attribute__((ext_vector_type(VEC)));
        const tv xv = *reinterpret_cast<const tv*>(xr + c);
        tv ov;
#pragma unroll
        for (int i = 0; i < VEC; ++i) ov[i] = (T)((float)xv[i] * alpha + (tr[c + i] + pr[c + i]));
        *reinterpret_cast<tv*>(orow + c) = ov;
      } else {
        for (int i = c; i < dim; ++i) orow[i] = (T)((float)xr[i] * alpha + (tr[i] + pr[i]));
      }
    }
  }
}

template <typename T>
static int fwd_launch(const void* x, void* out, const void* seq_offsets, const int64_t* timestamps, const void* num_targets,
                      const float* pos_w, const float* ts_w, int32_t* pos_idx, int32_t* ts_idx, int batch, int dim,
                      int max_ctx, int max_pos_ind, int max_bucket, int interleave, int fn, float alpha, int is64,
                      hipStream_t st) {
  hipLaunchKernelGGL((add_ts_pos_fwd_kernel<T>), dim3(batch), dim3(kPosThreads), 0, st, (const T*)x, (T*)out, seq_offsets,
                     timestamps, num_targets, pos_w, ts_w, pos_idx, ts_idx, dim, max_ctx, max_pos_ind, max_bucket, interleave,
                     fn, alpha, is64);
  return check_launch("hstu_add_ts_pos_emb_fwd");
}

}  // namespace hstu

using namespace hstu;

extern "C" {

int hstu_add_ts_pos_emb_fwd(const void* x, void* out, const void* seq_offsets, const int64_t* timestamps,
                            const void* num_targets, const float* pos_w, const float* ts_w, int32_t* pos_idx,
                            int32_t* ts_idx, int32_t batch, int32_t dim, int32_t max_contextual_seq_len,
                            int32_t max_pos_ind, int32_t max_time_bucket, int32_t interleave_targets,
                            int32_t time_bucket_fn, float alpha, int dtype, int index_dtype, void* stream) {
  if (!x || !out || !seq_offsets || !timestamps || !pos_w || !ts_w || !pos_idx || !ts_idx)
    return set_error(HSTU_EINVAL, "hstu_add_ts_pos_emb_fwd: x, out, seq_offsets, timestamps, tables and index outputs must be non-NULL");
  if (batch < 0 || dim <= 0 || max_pos_ind <= 0 || max_time_bucket < 0 || max_contextual_seq_len < 0)
    return set_error(HSTU_EINVAL, "hstu_add_ts_pos_emb_fwd: bad sizes");
  if (time_bucket_fn != 0 && time_bucket_fn != 1)
    return set_error(HSTU_EINVAL, "hstu_add_ts_pos_emb_fwd: time_bucket_fn must be 0 (sqrt) or 1 (log)");
  const int es = dtype == HSTU_DTYPE_F32 ? 4 : 2;
  if ((dim * es) % 16 || (((uintptr_t)x | (uintptr_t)out) & 15))
    return set_error(HSTU_EINVAL, "hstu_add_ts_pos_emb_fwd: rows must be 16-byte multiples and 16-byte aligned");
  if (batch == 0) return HSTU_OK;
  hipStream_t st = (hipStream_t)stream;
  const int is64 = index_dtype == HSTU_INDEX_I64;
#define CALL(T) fwd_launch<T>(x, out, seq_offsets, timestamps, num_targets, pos_w, ts_w, pos_idx, ts_idx, batch, dim, \
                              max_contextual_seq_len, max_pos_ind, max_time_bucket, interleave_targets, time_bucket_fn, alpha, is64, st)
  switch (dtype) {
    case HSTU_DTYPE_BF16: return CALL(bf16_t);
    case HSTU_DTYPE_F16: return CALL(f16_t);
    case HSTU_DTYPE_F32: return CALL(float);
    default: return set_error(HSTU_EINVAL, "dtype must be bf16, fp16 or fp32");
  }
#undef CALL
}

}  // extern "C"
