// Small kernels around the projections of an STU layer (ABI v10) and two calibration streams for bench.py.
//
//   hstu_cast_params   every parameter of a layer fp32 -> the activations' 16-bit dtype in ONE launch (the UVQK weight
//                      transposed to its K-contiguous copy on the way): replaces the six ``.to(x.dtype)`` kernels + one
//                      transposing copy per layer and forward (ops/hstu_compute.py:62-72, triton_hstu_linear.py:1160-1170
//                      cast them one by one).
//   hstu_column_sum    out[c] = sum over rows of x[r, c] in fp32, fixed summation order: the bias gradient of the UVQK
//                      projection (triton_addmm.py:309 ``torch.sum(dz, dim=0)``).  HBM-bound: one read of x.
//   hstu_calib_*       an MFMA stream without memory traffic and a read stream without arithmetic: what THIS box
//                      sustains, next to the product kernels' numbers in one bench line (box-to-box variance).
#include "capi_internal.h"
#include "hstu_common.cuh"

namespace hstu {

constexpr int kCastMaxItems = HSTU_CAST_MAX_ITEMS;

struct CastItems {
  const float* src[kCastMaxItems];
  void* dst[kCastMaxItems];
  int64_t numel[kCastMaxItems];
  int32_t rows[kCastMaxItems];    // transposed items: src is (rows, cols) row-major, dst (cols, rows)
  int32_t cols[kCastMaxItems];
  int32_t first_block[kCastMaxItems + 1];
  int32_t n;
};

template <typename T>
__global__ __launch_bounds__(256) void cast_params_kernel(const CastItems it) {
  __shared__ float tile[32][33];
  int item = 0;
#pragma unroll
  for (int i = 1; i < kCastMaxItems; ++i)
    if (i < it.n && (int)blockIdx.x >= it.first_block[i]) item = i;
  const int blk = blockIdx.x - it.first_block[item];
  const float* __restrict__ src = it.src[item];
  T* __restrict__ dst = (T*)it.dst[item];
  if (it.rows[item] == 0) {           // plain cast: 4 elements per thread
    const int64_t i0 = ((int64_t)blk * 256 + threadIdx.x) * 4;
    const int64_t n = it.numel[item];
    if (i0 + 3 < n && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
      const f32x4 v = *(const f32x4*)(src + i0);
      dst[i0] = (T)v[0]; dst[i0 + 1] = (T)v[1]; dst[i0 + 2] = (T)v[2]; dst[i0 + 3] = (T)v[3];
    } else {
      for (int e = 0; e < 4; ++e)
        if (i0 + e < n) dst[i0 + e] = (T)src[i0 + e];
    }
    return;
  }
  // transposing cast through a 32 x 32 LDS tile: coalesced reads along cols, coalesced writes along rows
  const int R = it.rows[item], Cc = it.cols[item];
  const int tiles_c = (Cc + 31) / 32;
  const int tr = blk / tiles_c, tc = blk % tiles_c;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int r = tr * 32 + ty + j, c = tc * 32 + tx;
    tile[ty + j][tx] = (r < R && c < Cc) ? src[(int64_t)r * Cc + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int c = tc * 32 + ty + j, r = tr * 32 + tx;
    if (r < R && c < Cc) dst[(int64_t)c * R + r] = (T)tile[tx][ty + j];
  }
}

// the two 16-bit values of a dword as fp32
template <typename T> struct Pair16;
template <> struct Pair16<bf16_t> {
  static HSTU_DEV float lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
  static HSTU_DEV float hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
};
template <> struct Pair16<f16_t> {
  typedef f16_t v2 __attribute__((ext_vector_type(2)));
  static HSTU_DEV float lo(uint32_t w) { return (float)__builtin_bit_cast(v2, w)[0]; }
  static HSTU_DEV float hi(uint32_t w) { return (float)__builtin_bit_cast(v2, w)[1]; }
};

// ---- column sums: thread = 8 consecutive 16-bit columns (16 bytes), rows dealt round-robin to the row groups of the grid;
// partial[group][cols] fp32, then a second kernel adds the groups in index order (deterministic)
constexpr int kColSumGroupsMax = 1024;

template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void column_sum_kernel(const T* __restrict__ x, int64_t ldx, int64_t rows, int cols,
                                                           float* __restrict__ partial) {
  const int c0 = (blockIdx.y * 256 + threadIdx.x) * 8;
  if (c0 >= cols) return;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int64_t g = blockIdx.x, G = gridDim.x;
  int64_t r = g;
  for (; r + (UNROLL - 1) * G < rows; r += UNROLL * G) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = gload16(x + (r + u * G) * ldx + c0);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += Pair16<T>::lo(v[u][e]);
        acc[2 * e + 1] += Pair16<T>::hi(v[u][e]);
      }
  }
  for (; r < rows; r += G) {
    const u32x4 v = gload16(x + r * ldx + c0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[2 * e] += Pair16<T>::lo(v[e]);
      acc[2 * e + 1] += Pair16<T>::hi(v[e]);
    }
  }
  float* dst = partial + g * cols + c0;
  *(f32x4*)dst = f32x4{acc[0], acc[1], acc[2], acc[3]};
  *(f32x4*)(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

// partial (groups, cols) -> out (cols): block = 16 columns x 16 group sixteenths, each thread adds its share of the partials in group
// order (four independent running sums), the sixteen shares are added in index order through LDS: a fixed summation order
__global__ __launch_bounds__(256) void column_sum_finish_kernel(const float* __restrict__ partial, int groups, int cols,
                                                                 float* __restrict__ out) {
  __shared__ float part[16][17];
  const int cl = threadIdx.x & 15, sh = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int per = (groups + 15) / 16;
  const int g0 = sh * per, g1 = min(g0 + per, groups);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int g = g0;
    for (; g + 3 < g1; g += 4) {
      s0 += partial[(int64_t)g * cols + c];
      s1 += partial[(int64_t)(g + 1) * cols + c];
      s2 += partial[(int64_t)(g + 2) * cols + c];
      s3 += partial[(int64_t)(g + 3) * cols + c];
    }
    for (; g < g1; ++g) s0 += partial[(int64_t)g * cols + c];
  }
  part[sh][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sh == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][cl];
    out[c] = t;
  }
}

static int column_sum_groups(int64_t rows, int cols) {
  const int n_cu = cu_count();
  const int col_blocks = (cols / 8 + 255) / 256;
  int64_t g = (int64_t)n_cu * 3 / col_blocks;       // 3 workgroups of 4 waves per CU, 8 rows of 16 bytes per lane in flight
  if (g > rows) g = rows;
  if (g > kColSumGroupsMax) g = kColSumGroupsMax;
  return g < 1 ? 1 : (int)g;
}

// ---- calibration streams
template <int DUMMY>
__global__ __launch_bounds__(256) void calib_mfma_kernel(int iters, float* sink) {
  typedef __bf16 b8 __attribute__((ext_vector_type(8)));
  b8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x ^ e)); }
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) sink[0] = s;     // never true: keeps the chain alive
}

__global__ __launch_bounds__(256) void calib_read_kernel(const u32x4* __restrict__ src, int64_t n16, float* sink) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u32x4 v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + stride);
    const u32x4 v2 = __builtin_nontemporal_load(src + i + 2 * stride), v3 = __builtin_nontemporal_load(src + i + 3 * stride);
    acc ^= v0 ^ v1 ^ v2 ^ v3;
  }
  for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(src + i);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) sink[0] = 1.f;   // practically never: keeps the loads alive
}

}  // namespace hstu

using namespace hstu;

extern "C" {

int hstu_cast_params(const HstuCastItem* items, int32_t n_items, int dst_dtype, void* stream) {
  if (n_items == 0) return HSTU_OK;
  if (!items || n_items < 0 || n_items > kCastMaxItems) return set_error(HSTU_EINVAL, "hstu_cast_params: 1..%d items", kCastMaxItems);
  if (dst_dtype != HSTU_DTYPE_BF16 && dst_dtype != HSTU_DTYPE_F16) return set_error(HSTU_EINVAL, "hstu_cast_params: the destination dtype must be bf16 or fp16");
  CastItems it;
  it.n = n_items;
  int blocks = 0;
  for (int i = 0; i < n_items; ++i) {
    const HstuCastItem& s = items[i];
    if (!s.src || !s.dst || s.numel < 0) return set_error(HSTU_EINVAL, "hstu_cast_params: item %d: NULL tensor / negative size", i);
    if (s.transpose && (s.rows <= 0 || s.cols <= 0 || (int64_t)s.rows * s.cols != s.numel))
      return set_error(HSTU_EINVAL, "hstu_cast_params: item %d: rows x cols must equal numel for a transposing cast", i);
    it.src[i] = s.src; it.dst[i] = s.dst; it.numel[i] = s.numel;
    it.rows[i] = s.transpose ? s.rows : 0;
    it.cols[i] = s.transpose ? s.cols : 0;
    it.first_block[i] = blocks;
    const int64_t nb = s.transpose ? (int64_t)((s.rows + 31) / 32) * ((s.cols + 31) / 32) : (s.numel + 1023) / 1024;
    if (blocks + nb > (1 << 30)) return set_error(HSTU_EINVAL, "hstu_cast_params: too many elements");
    blocks += (int)nb;
  }
  for (int i = n_items; i <= kCastMaxItems; ++i) it.first_block[i] = blocks;
  if (blocks == 0) return HSTU_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dst_dtype == HSTU_DTYPE_BF16) hipLaunchKernelGGL(cast_params_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, it);
  else hipLaunchKernelGGL(cast_params_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, it);
  return check_launch("hstu_cast_params");
}

size_t hstu_column_sum_workspace_bytes(int64_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return (size_t)column_sum_groups(rows, cols) * cols * sizeof(float);
}

int hstu_column_sum(const void* x, int64_t ldx, int64_t rows, int32_t cols, float* out, void* workspace, int dtype, void* stream) {
  if (cols <= 0 || rows < 0) return set_error(HSTU_EINVAL, "hstu_column_sum: bad shape");
  if (!out) return set_error(HSTU_EINVAL, "hstu_column_sum: out is NULL");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), st);
    return e == hipSuccess ? HSTU_OK : set_error(HSTU_ELAUNCH, "hstu_column_sum: memset failed: %s", hipGetErrorString(e));
  }
  if (dtype != HSTU_DTYPE_BF16 && dtype != HSTU_DTYPE_F16) return set_error(HSTU_EUNSUPPORTED, "hstu_column_sum: bf16 / fp16 rows only");
  if (!x || !workspace) return set_error(HSTU_EINVAL, "hstu_column_sum: x / workspace is NULL");
  if (cols % 8 || ldx % 8 || ldx < cols || ((uintptr_t)x & 15) || ((uintptr_t)workspace & 15))
    return set_error(HSTU_EINVAL, "hstu_column_sum: cols and the leading dimension must be multiples of 8, pointers 16-byte aligned");
  const int groups = column_sum_groups(rows, cols);
  const dim3 grid(groups, (cols / 8 + 255) / 256);
  float* partial = (float*)workspace;
  if (dtype == HSTU_DTYPE_BF16) hipLaunchKernelGGL((column_sum_kernel<bf16_t, 8>), grid, dim3(256), 0, st, (const bf16_t*)x, ldx, rows, cols, partial);
  else hipLaunchKernelGGL((column_sum_kernel<f16_t, 8>), grid, dim3(256), 0, st, (const f16_t*)x, ldx, rows, cols, partial);
  if (int e = check_launch("hstu_column_sum")) return e;
  hipLaunchKernelGGL(column_sum_finish_kernel, dim3((cols + 15) / 16), dim3(256), 0, st, partial, groups, cols, out);
  return check_launch("hstu_column_sum(finish)");
}

int hstu_calib_mfma_stream(int32_t iters, float* sink, double* flops, void* stream) {
  if (iters <= 0 || !sink) return set_error(HSTU_EINVAL, "hstu_calib_mfma_stream: iters > 0 and a sink are required");
  const int n_cu = cu_count();
  const int blocks = n_cu * 2;          // 2 workgroups x 4 waves per CU: two waves per SIMD, as the product's MFMA kernels
  hipLaunchKernelGGL(calib_mfma_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
  if (flops) *flops = (double)blocks * 4 * (double)iters * 16 * (2.0 * 32 * 32 * 16);
  return check_launch("hstu_calib_mfma_stream");
}

int hstu_calib_read_stream(const void* src, size_t bytes, float* sink, void* stream) {
  if (!src || !sink || ((uintptr_t)src & 15)) return set_error(HSTU_EINVAL, "hstu_calib_read_stream: a 16-byte aligned source and a sink are required");
  const int n_cu = cu_count();
  hipLaunchKernelGGL(calib_read_kernel, dim3(n_cu * 8), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (int64_t)(bytes / 16), sink);
  return check_launch("hstu_calib_read_stream");
}

}  // extern "C"
