// Jagged-tensor helpers for gfx950: pure index arithmetic + byte copies (bit-exact).
// HBM-bound: every payload byte is read once and written once with the widest vector
// the row size allows (16 B per lane when rows are multiples of 16 B); one workgroup
// walks a chunk of rows of ONE user so offset loads are scalar and amortised.
//
// Reference semantics: ops/pytorch/pt_jagged_tensors.py:31-246 (concat / split and the
// l2-embedding prefix variants), ops/triton/triton_jagged_tensors.py:31-142 (kernels we
// replace), fbgemm jagged_to_padded_dense / dense_to_jagged / asynchronous_complete_cumsum
// (call sites ops/pytorch/pt_hstu_attention.py:97-171, modules/stu.py:97),
// ops/cpp/{complete_cumsum,expand_1d_jagged_to_dense,concat_1d_jagged_jagged}.cu.
#include "hstu_common.cuh"
#include "capi_internal.h"

#ifndef JAGGED_NT
#define JAGGED_NT 0     // non-temporal hint on the byte movers' loads (1) / stores (2)
#endif

namespace hstu {

constexpr int kRowsPerBlock = 16;
constexpr int kCopyThreads = 256;

template <int V> struct VecT;
template <> struct VecT<16> { typedef u32x4 type; };
template <> struct VecT<8> { typedef u32x2 type; };
template <> struct VecT<4> { typedef uint32_t type; };
template <> struct VecT<2> { typedef uint16_t type; };
template <> struct VecT<1> { typedef uint8_t type; };

// copy (or zero-fill when src == nullptr) `row_bytes` bytes, cooperatively by `nthr` threads
template <int V>
HSTU_DEV void copy_row(char* dst, const char* src, int row_bytes, int t, int nthr) {
  typedef typename VecT<V>::type vt;
  const int n = row_bytes / V;
  for (int i = t; i < n; i += nthr) {
    vt x = !src ? vt{} : (JAGGED_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const vt*>(src) + i) : reinterpret_cast<const vt*>(src)[i];
    if (JAGGED_NT & 2) __builtin_nontemporal_store(x, reinterpret_cast<vt*>(dst) + i);
    else reinterpret_cast<vt*>(dst)[i] = x;
  }
}

HSTU_DEV int64_t side_offset(const void* offsets, int b, int max_len, int is64) {
  return offsets ? load_index(offsets, b, is64) : (int64_t)b * max_len;
}

// grid = (B, ceil(max_seq_len / kRowsPerBlock))
// SPLIT == false: out[b] = [right[:np] ; left ; right[np:]]   (concat)
// SPLIT == true : the inverse scatter
template <int V, bool SPLIT>
__global__ __launch_bounds__(kCopyThreads) void concat_split_kernel(char* left, char* right, char* comb,
                                                                    const void* off_l, const void* off_r,
                                                                    int max_len_l, int max_len_r, int row_bytes,
                                                                    int n_prefix, int is64) {
  const int b = blockIdx.x;   // users on grid.x (2^31 - 1 blocks); grid.y is limited to 65535
  const int64_t ol = side_offset(off_l, b, max_len_l, is64);
  const int64_t orr = side_offset(off_r, b, max_len_r, is64);
  const int ll = (int)(side_offset(off_l, b + 1, max_len_l, is64) - ol);
  const int lr = (int)(side_offset(off_r, b + 1, max_len_r, is64) - orr);
  const int total = ll + lr;
  const int r_begin = blockIdx.y * kRowsPerBlock;
  if (r_begin >= total) return;
  const int np = min(n_prefix, lr);
  // threads split as (rows in flight) x (lanes per row)
  const int units = max(row_bytes / V, 1);
  int tpr = 1;
  while (tpr < units && tpr < kCopyThreads) tpr <<= 1;       // threads per row (power of two)
  const int rows_par = kCopyThreads / tpr;
  const int t_in_row = threadIdx.x % tpr, my_row_slot = threadIdx.x / tpr;
  const int r_end = min(r_begin + kRowsPerBlock, total);
  for (int r = r_begin + my_row_slot; r < r_end; r += rows_par) {
    char* side;
    if (r < np) side = right + (orr + r) * (int64_t)row_bytes;
    else if (r < np + ll) side = left + (ol + (r - np)) * (int64_t)row_bytes;
    else side = right + (orr + (r - ll)) * (int64_t)row_bytes;
    char* c = comb + (ol + orr + r) * (int64_t)row_bytes;
    if (SPLIT) copy_row<V>(side, c, row_bytes, t_in_row, tpr);
    else copy_row<V>(c, side, row_bytes, t_in_row, tpr);
  }
}

// grid = (ceil(max_len / kRowsPerBlock), B).  TO_DENSE: values -> dense (+ zero fill); else dense -> values
template <int V, bool TO_DENSE>
__global__ __launch_bounds__(kCopyThreads) void padded_dense_kernel(char* values, char* dense, const void* offsets,
                                                                    int max_len, int row_bytes, int is64) {
  const int b = blockIdx.x;   // users on grid.x (2^31 - 1 blocks); grid.y is limited to 65535
  const int64_t off = load_index(offsets, b, is64);
  const int len = min((int)(load_index(offsets, b + 1, is64) - off), max_len);
  const int r_begin = blockIdx.y * kRowsPerBlock;
  const int limit = TO_DENSE ? max_len : len;
  if (r_begin >= limit) return;
  const int units = max(row_bytes / V, 1);
  int tpr = 1;
  while (tpr < units && tpr < kCopyThreads) tpr <<= 1;
  const int rows_par = kCopyThreads / tpr;
  const int t_in_row = threadIdx.x % tpr, my_row_slot = threadIdx.x / tpr;
  const int r_end = min(r_begin + kRowsPerBlock, limit);
  for (int r = r_begin + my_row_slot; r < r_end; r += rows_par) {
    char* d = dense + ((int64_t)b * max_len + r) * row_bytes;
    char* v = values + (off + r) * (int64_t)row_bytes;
    if (TO_DENSE) copy_row<V>(d, r < len ? v : nullptr, row_bytes, t_in_row, tpr);
    else copy_row<V>(v, d, row_bytes, t_in_row, tpr);
  }
}

// grid = (ceil(tail / kRowsPerBlock), B): the LAST `tail` rows of every user's region of a jagged buffer are
// overwritten with the user's `tail` dense rows (the in-place KV-cache append: O(tail), not O(history))
template <int V>
__global__ __launch_bounds__(kCopyThreads) void write_tail_kernel(const char* dense, char* values, const void* offsets,
                                                                  int tail, int row_bytes, int is64) {
  const int b = blockIdx.x;   // users on grid.x (2^31 - 1 blocks); grid.y is limited to 65535
  const int64_t end = load_index(offsets, b + 1, is64);
  const int r_begin = blockIdx.y * kRowsPerBlock;
  if (r_begin >= tail) return;
  const int units = max(row_bytes / V, 1);
  int tpr = 1;
  while (tpr < units && tpr < kCopyThreads) tpr <<= 1;
  const int rows_par = kCopyThreads / tpr;
  const int t_in_row = threadIdx.x % tpr, my_row_slot = threadIdx.x / tpr;
  const int r_end = min(r_begin + kRowsPerBlock, tail);
  for (int r = r_begin + my_row_slot; r < r_end; r += rows_par)
    copy_row<V>(values + (end - tail + r) * (int64_t)row_bytes, dense + ((int64_t)b * tail + r) * row_bytes, row_bytes,
                t_in_row, tpr);
}

// out[0] = 0, out[i+1] = in[0] + ... + in[i].  One workgroup, chunked wave scan with carry
// (B is at most a few 100k; the op is latency- not bandwidth-bound).
template <typename I>
__global__ __launch_bounds__(1024) void complete_cumsum_kernel(const I* in, I* out, int64_t n) {
  __shared__ I wave_tot[16];
  __shared__ I carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { out[0] = 0; carry_s = 0; }
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + tid;
    I x = i < n ? in[i] : (I)0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      I y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    if (lane == 63) wave_tot[wave] = x;
    __syncthreads();
    I pre = carry_s;
    for (int w = 0; w < wave; ++w) pre += wave_tot[w];
    if (i < n) out[i + 1] = x + pre;
    __syncthreads();
    if (tid == 1023) carry_s = x + pre;
    __syncthreads();
  }
}

// (B, max_len) <- 1-D jagged, padded with the user's last value (0 when empty)
template <typename E>
__global__ void expand_1d_kernel(const E* values, const void* offsets, E* dense, int max_len, int is64) {
  const int b = blockIdx.x;   // users on grid.x (2^31 - 1 blocks); grid.y is limited to 65535
  const int64_t off = load_index(offsets, b, is64);
  const int len = (int)(load_index(offsets, b + 1, is64) - off);
  const int i = blockIdx.y * blockDim.x + threadIdx.x;
  if (i >= max_len) return;
  E x = (E)0;
  if (len > 0) x = values[off + min(i, len - 1)];
  dense[(int64_t)b * max_len + i] = x;
}

template <typename E>
__global__ void concat_1d_kernel(const E* vl, const void* ol, const E* vr, const void* orr, E* out, int is64) {
  const int b = blockIdx.x;
  const int64_t a0 = load_index(ol, b, is64), b0 = load_index(orr, b, is64);
  const int la = (int)(load_index(ol, b + 1, is64) - a0), lb = (int)(load_index(orr, b + 1, is64) - b0);
  for (int i = threadIdx.x; i < la + lb; i += blockDim.x) out[a0 + b0 + i] = i < la ? vl[a0 + i] : vr[b0 + i - la];
}

static int pick_vec(int row_bytes, const void* a, const void* b, const void* c) {
  uintptr_t bits = (uintptr_t)row_bytes | (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
  if ((bits & 15) == 0) return 16;
  if ((bits & 7) == 0) return 8;
  if ((bits & 3) == 0) return 4;
  if ((bits & 1) == 0) return 2;
  return 1;
}

template <bool SPLIT>
static int launch_concat_split(char* left, char* right, char* comb, const void* off_l, const void* off_r,
                               int max_len_l, int max_len_r, int max_seq_len, int batch, int row_bytes, int n_prefix,
                               int is64, hipStream_t st) {
  if (batch == 0 || row_bytes == 0 || max_seq_len == 0) return HSTU_OK;
  dim3 grid(batch, (max_seq_len + kRowsPerBlock - 1) / kRowsPerBlock);
  const int v = pick_vec(row_bytes, left, right, comb);
#define LAUNCH(V)                                                                                              \
  hipLaunchKernelGGL((concat_split_kernel<V, SPLIT>), grid, dim3(kCopyThreads), 0, st, left, right, comb, off_l, \
                     off_r, max_len_l, max_len_r, row_bytes, n_prefix, is64)
  switch (v) {
    case 16: LAUNCH(16); break;
    case 8: LAUNCH(8); break;
    case 4: LAUNCH(4); break;
    case 2: LAUNCH(2); break;
    default: LAUNCH(1); break;
  }
#undef LAUNCH
  return check_launch("concat/split_2d_jagged");
}

template <bool TO_DENSE>
static int launch_padded(char* values, char* dense, const void* offsets, int batch, int max_len, int row_bytes,
                         int is64, hipStream_t st) {
  if (batch == 0 || row_bytes == 0 || max_len == 0) return HSTU_OK;
  dim3 grid(batch, (max_len + kRowsPerBlock - 1) / kRowsPerBlock);
  const int v = pick_vec(row_bytes, values, dense, nullptr);
#define LAUNCH(V)                                                                                                 \
  hipLaunchKernelGGL((padded_dense_kernel<V, TO_DENSE>), grid, dim3(kCopyThreads), 0, st, values, dense, offsets, \
                     max_len, row_bytes, is64)
  switch (v) {
    case 16: LAUNCH(16); break;
    case 8: LAUNCH(8); break;
    case 4: LAUNCH(4); break;
    case 2: LAUNCH(2); break;
    default: LAUNCH(1); break;
  }
#undef LAUNCH
  return check_launch("jagged<->padded_dense");
}

}  // namespace hstu

using namespace hstu;

extern "C" {

int hstu_complete_cumsum(const void* in, void* out, int64_t n, int index_dtype, void* stream) {
  if (n < 0 || !out || (n > 0 && !in)) return set_error(HSTU_EINVAL, "complete_cumsum: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (index_dtype == HSTU_INDEX_I64)
    hipLaunchKernelGGL(complete_cumsum_kernel<int64_t>, dim3(1), dim3(1024), 0, st, (const int64_t*)in, (int64_t*)out, n);
  else
    hipLaunchKernelGGL(complete_cumsum_kernel<int32_t>, dim3(1), dim3(1024), 0, st, (const int32_t*)in, (int32_t*)out, n);
  return check_launch("complete_cumsum");
}

int hstu_concat_2d_jagged(const void* left, const void* right, void* out, const void* offsets_left,
                          const void* offsets_right, int32_t max_len_left, int32_t max_len_right, int32_t max_seq_len,
                          int32_t batch, int32_t dim, int32_t elem_bytes, int32_t n_prefix, int index_dtype,
                          void* stream) {
  if (!offsets_left && !offsets_right && (max_len_left <= 0 && max_len_right <= 0))
    return set_error(HSTU_EINVAL, "concat_2d_jagged: a side without offsets needs its max_len");
  return launch_concat_split<false>((char*)left, (char*)right, (char*)out, offsets_left, offsets_right, max_len_left,
                                    max_len_right, max_seq_len, batch, dim * elem_bytes, n_prefix,
                                    index_dtype == HSTU_INDEX_I64, (hipStream_t)stream);
}

int hstu_split_2d_jagged(const void* in, void* left, void* right, const void* offsets_left, const void* offsets_right,
                         int32_t max_len_left, int32_t max_len_right, int32_t max_seq_len, int32_t batch, int32_t dim,
                         int32_t elem_bytes, int32_t n_prefix, int index_dtype, void* stream) {
  if (!offsets_left && !offsets_right)
    return set_error(HSTU_EINVAL, "split_2d_jagged: offsets_left and offsets_right cannot both be NULL");
  return launch_concat_split<true>((char*)left, (char*)right, (char*)in, offsets_left, offsets_right, max_len_left,
                                   max_len_right, max_seq_len, batch, dim * elem_bytes, n_prefix,
                                   index_dtype == HSTU_INDEX_I64, (hipStream_t)stream);
}

int hstu_jagged_to_padded_dense(const void* values, void* dense, const void* offsets, int32_t batch, int32_t max_len,
                                int32_t dim, int32_t elem_bytes, int index_dtype, void* stream) {
  if (!offsets) return set_error(HSTU_EINVAL, "jagged_to_padded_dense: offsets is NULL");
  return launch_padded<true>((char*)values, (char*)dense, offsets, batch, max_len, dim * elem_bytes,
                             index_dtype == HSTU_INDEX_I64, (hipStream_t)stream);
}

int hstu_dense_to_jagged(const void* dense, void* values, const void* offsets, int32_t batch, int32_t max_len,
                         int32_t dim, int32_t elem_bytes, int index_dtype, void* stream) {
  if (!offsets) return set_error(HSTU_EINVAL, "dense_to_jagged: offsets is NULL");
  return launch_padded<false>((char*)values, (char*)dense, offsets, batch, max_len, dim * elem_bytes,
                              index_dtype == HSTU_INDEX_I64, (hipStream_t)stream);
}

int hstu_jagged_write_tail(const void* dense, void* values, const void* offsets, int32_t batch, int32_t tail, int32_t dim,
                           int32_t elem_bytes, int index_dtype, void* stream) {
  if (batch == 0 || tail == 0 || dim == 0) return HSTU_OK;
  if (!dense || !values || !offsets) return set_error(HSTU_EINVAL, "jagged_write_tail: NULL tensor");
  if (batch < 0 || tail < 0 || dim < 0) return set_error(HSTU_EINVAL, "jagged_write_tail: negative size");
  const int row_bytes = dim * elem_bytes;
  const int is64 = index_dtype == HSTU_INDEX_I64;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(batch, (tail + kRowsPerBlock - 1) / kRowsPerBlock);
#define LAUNCH(V) hipLaunchKernelGGL((write_tail_kernel<V>), grid, dim3(kCopyThreads), 0, st, (const char*)dense, (char*)values, offsets, tail, row_bytes, is64)
  switch (pick_vec(row_bytes, dense, values, nullptr)) {
    case 16: LAUNCH(16); break;
    case 8: LAUNCH(8); break;
    case 4: LAUNCH(4); break;
    case 2: LAUNCH(2); break;
    default: LAUNCH(1); break;
  }
#undef LAUNCH
  return check_launch("jagged_write_tail");
}

int hstu_expand_1d_jagged_to_dense(const void* values, const void* offsets, void* dense, int32_t batch, int32_t max_len,
                                   int32_t elem_bytes, int index_dtype, void* stream) {
  if (batch == 0 || max_len == 0) return HSTU_OK;
  if (elem_bytes != 4 && elem_bytes != 8) return set_error(HSTU_EINVAL, "expand_1d_jagged_to_dense: elem_bytes must be 4 or 8");
  dim3 grid(batch, (max_len + 63) / 64);
  const int is64 = index_dtype == HSTU_INDEX_I64;
  if (elem_bytes == 8)
    hipLaunchKernelGGL(expand_1d_kernel<int64_t>, grid, dim3(64), 0, (hipStream_t)stream, (const int64_t*)values, offsets,
                       (int64_t*)dense, max_len, is64);
  else
    hipLaunchKernelGGL(expand_1d_kernel<int32_t>, grid, dim3(64), 0, (hipStream_t)stream, (const int32_t*)values, offsets,
                       (int32_t*)dense, max_len, is64);
  return check_launch("expand_1d_jagged_to_dense");
}

int hstu_concat_1d_jagged_jagged(const void* values_left, const void* offsets_left, const void* values_right,
                                 const void* offsets_right, void* out, int32_t batch, int32_t elem_bytes,
                                 int index_dtype, void* stream) {
  if (batch == 0) return HSTU_OK;
  if (elem_bytes != 4 && elem_bytes != 8) return set_error(HSTU_EINVAL, "concat_1d_jagged_jagged: elem_bytes must be 4 or 8");
  const int is64 = index_dtype == HSTU_INDEX_I64;
  if (elem_bytes == 8)
    hipLaunchKernelGGL(concat_1d_kernel<int64_t>, dim3(batch), dim3(128), 0, (hipStream_t)stream, (const int64_t*)values_left,
                       offsets_left, (const int64_t*)values_right, offsets_right, (int64_t*)out, is64);
  else
    hipLaunchKernelGGL(concat_1d_kernel<int32_t>, dim3(batch), dim3(128), 0, (hipStream_t)stream, (const int32_t*)values_left,
                       offsets_left, (const int32_t*)values_right, offsets_right, (int32_t*)out, is64);
  return check_launch("concat_1d_jagged_jagged");
}

}  // extern "C"
