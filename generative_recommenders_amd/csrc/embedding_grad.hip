// Table gradients of the timestamp / position encoder: table_grad[i] = sum of the dout rows whose
// table index is i -- the index_select backward of one embedding table over 10^5..10^6 jagged rows and <= 8 K table rows.
// Kernels replaced: _add_embeddings_bwd_kernel (ops/triton/triton_position.py:188-238) and the sort on its host side
// (:339-407).
//
// 1. rows are grouped by table index with a radix sort over only the ceil(log2(table_rows)) significant key bits
//    (rocPRIM device sort, 32-bit keys and 32-bit row numbers from a counting iterator: two digit passes for an 8 K-row
//    table, against the eight of a full-width sort with int64 payload);
// 2. one wave per run of 64 sorted rows: the 64 (index, row) pairs arrive with one coalesced load and are handed out
//    lane by lane; the wave reads four rows at a time (16 bytes per lane and piece), adds them IN SORTED ORDER into fp32
//    column sums in registers and flushes when the table index changes: a segment inside one run is written by exactly
//    one wave with a plain store, segments cut by a run boundary add their pieces with fp32 atomics (the only
//    non-deterministic summation order, and only for those rows).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "hstu_common.cuh"
#include "capi_internal.h"

namespace hstu {

constexpr int kSegWaves = 4;
constexpr int kSegRun = 64;       // sorted rows per wave
constexpr int kSegPieces = 4;     // 16-byte pieces per lane and row: rows of up to 4 KiB

template <typename T, typename P>
__global__ __launch_bounds__(kSegWaves * 64) void segment_sum_kernel(const T* __restrict__ g, const P* __restrict__ perm,
                                                                     const int32_t* __restrict__ sorted_idx, int64_t n, int dim,
                                                                     float* __restrict__ table) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t e0 = ((int64_t)blockIdx.x * kSegWaves + wave) * kSegRun;
  if (e0 >= n) return;
  const int cnt = (int)min((int64_t)kSegRun, n - e0);
  const int my_idx = lane < cnt ? sorted_idx[e0 + lane] : -1;
  const int64_t my_row = lane < cnt ? (int64_t)perm[e0 + lane] : 0;
  const int pieces = dim / VEC;                     // dim * sizeof(T) is a multiple of 16 (checked by the caller)
  float acc[kSegPieces][VEC];
#pragma unroll
  for (int j = 0; j < kSegPieces; ++j)
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[j][c] = 0.f;
  int cur = __shfl(my_idx, 0);
  bool shared = e0 > 0 && sorted_idx[e0 - 1] == cur;      // the first segment started in the previous run
  auto flush = [&](int idx, bool atomic) {
#pragma unroll
    for (int j = 0; j < kSegPieces; ++j) {
      const int p = lane + 64 * j;
      if (p < pieces) {
        float* dst = table + (int64_t)idx * dim + p * VEC;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          if (atomic) atomicAdd(dst + c, acc[j][c]);
          else dst[c] = acc[j][c];
          acc[j][c] = 0.f;
        }
      }
    }
  };
  for (int r0 = 0; r0 < cnt; r0 += 4) {
    u32x4 v[4][kSegPieces];
    int idx4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = min(r0 + u, cnt - 1);
      idx4[u] = __shfl(my_idx, r);
      const char* row = (const char*)(g + __shfl(my_row, r) * dim);
#pragma unroll
      for (int j = 0; j < kSegPieces; ++j) {
        const int p = lane + 64 * j;
        if (p < pieces) v[u][j] = *(const u32x4*)(row + 16 * p);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r0 + u >= cnt) break;
      if (idx4[u] != cur) {
        flush(cur, shared);
        cur = idx4[u];
        shared = false;
      }
#pragma unroll
      for (int j = 0; j < kSegPieces; ++j) {
        if (lane + 64 * j < pieces) {
          const T* e = (const T*)&v[u][j];
#pragma unroll
          for (int c = 0; c < VEC; ++c) acc[j][c] += (float)e[c];
        }
      }
    }
  }
  const bool open_right = e0 + cnt < n && sorted_idx[e0 + cnt] == cur;
  flush(cur, shared || open_right);
}

template <typename T, typename P>
static int seg_launch(const void* g, const P* perm, const int32_t* sorted_idx, int64_t n, int dim, float* table, hipStream_t st,
                      const char* what) {
  const int blocks = (int)((n + kSegWaves * kSegRun - 1) / (kSegWaves * kSegRun));
  hipLaunchKernelGGL((segment_sum_kernel<T, P>), dim3(blocks), dim3(kSegWaves * 64), 0, st, (const T*)g, perm, sorted_idx, n, dim,
                     table);
  return check_launch(what);
}

template <typename P>
static int seg_dispatch(const void* g, const P* perm, const int32_t* sorted_idx, int64_t n, int dim, float* table, int dtype,
                        hipStream_t st, const char* what) {
  switch (dtype) {
    case HSTU_DTYPE_BF16: return seg_launch<bf16_t, P>(g, perm, sorted_idx, n, dim, table, st, what);
    case HSTU_DTYPE_F16: return seg_launch<f16_t, P>(g, perm, sorted_idx, n, dim, table, st, what);
    case HSTU_DTYPE_F32: return seg_launch<float, P>(g, perm, sorted_idx, n, dim, table, st, what);
    default: return set_error(HSTU_EINVAL, "dtype must be bf16, fp16 or fp32");
  }
}

static int seg_check(const char* what, const void* dout, int64_t n, int dim, int table_rows, const float* table_grad, int dtype) {
  if (!table_grad || table_rows <= 0 || dim <= 0) return set_error(HSTU_EINVAL, "%s: table_grad must be non-NULL, sizes positive", what);
  const int es = dtype == HSTU_DTYPE_F32 ? 4 : 2;
  if ((dim * es) % 16 || ((uintptr_t)dout & 15)) return set_error(HSTU_EINVAL, "%s: rows must be 16-byte multiples and 16-byte aligned", what);
  if (dim * es > kSegPieces * 64 * 16) return set_error(HSTU_EUNSUPPORTED, "%s: rows of %d bytes > %d", what, dim * es, kSegPieces * 64 * 16);
  if (n < 0 || n > 0x7fffffffLL) return set_error(HSTU_EINVAL, "%s: n out of range", what);
  return HSTU_OK;
}

static unsigned key_bits(int table_rows) {
  unsigned bits = 1;
  while ((1u << bits) < (unsigned)table_rows && bits < 31) ++bits;
  return bits;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sort_temp_bytes(int64_t n, int table_rows) {
  size_t bytes = 0;
  rocprim::counting_iterator<int32_t> rows(0);
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, rows, (int32_t*)nullptr, (size_t)n, 0u,
                                  key_bits(table_rows), (hipStream_t)0);
  return bytes;
}

}  // namespace hstu

using namespace hstu;

extern "C" {

int hstu_embedding_grad_segment_sum(const void* dout, const int64_t* sorted_rows, const int32_t* sorted_idx, int64_t n,
                                    int32_t dim, int32_t table_rows, float* table_grad, int dtype, void* stream) {
  const char* what = "hstu_embedding_grad_segment_sum";
  if (int rc = seg_check(what, dout, n, dim, table_rows, table_grad, dtype)) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(table_grad, 0, (size_t)table_rows * dim * sizeof(float), st);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "%s: memset failed: %s", what, hipGetErrorString(e));
  if (n == 0) return HSTU_OK;
  if (!dout || !sorted_rows || !sorted_idx) return set_error(HSTU_EINVAL, "%s: NULL input", what);
  return seg_dispatch<int64_t>(dout, sorted_rows, sorted_idx, n, dim, table_grad, dtype, st, what);
}

int hstu_embedding_grad_workspace_bytes(int64_t n, int32_t table_rows, int64_t* bytes) {
  if (!bytes || n < 0 || n > 0x7fffffffLL || table_rows <= 0) return set_error(HSTU_EINVAL, "hstu_embedding_grad_workspace_bytes: bad arguments");
  *bytes = (int64_t)(2 * align256((size_t)n * sizeof(int32_t)) + align256(n ? sort_temp_bytes(n, table_rows) : 0));
  return HSTU_OK;
}

int hstu_embedding_grad(const void* dout, const int32_t* idx, int64_t n, int32_t dim, int32_t table_rows, float* table_grad,
                        void* workspace, int64_t workspace_bytes, int dtype, void* stream) {
  const char* what = "hstu_embedding_grad";
  if (int rc = seg_check(what, dout, n, dim, table_rows, table_grad, dtype)) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(table_grad, 0, (size_t)table_rows * dim * sizeof(float), st);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "%s: memset failed: %s", what, hipGetErrorString(e));
  if (n == 0) return HSTU_OK;
  if (!dout || !idx || !workspace) return set_error(HSTU_EINVAL, "%s: NULL input", what);
  const size_t part = align256((size_t)n * sizeof(int32_t));
  size_t temp = sort_temp_bytes(n, table_rows);
  if (workspace_bytes < (int64_t)(2 * part + align256(temp)) || ((uintptr_t)workspace & 255))
    return set_error(HSTU_EINVAL, "%s: workspace too small or not 256-byte aligned (hstu_embedding_grad_workspace_bytes)", what);
  int32_t* sorted_idx = (int32_t*)workspace;
  int32_t* perm = (int32_t*)((char*)workspace + part);
  void* tmp = (char*)workspace + 2 * part;
  rocprim::counting_iterator<int32_t> rows(0);
  e = rocprim::radix_sort_pairs(tmp, temp, idx, sorted_idx, rows, perm, (size_t)n, 0u, key_bits(table_rows), st);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "%s: sort failed: %s", what, hipGetErrorString(e));
  return seg_dispatch<int32_t>(dout, perm, sorted_idx, n, dim, table_grad, dtype, st, what);
}

}  // extern "C"
