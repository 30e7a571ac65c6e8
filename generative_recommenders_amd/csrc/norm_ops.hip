// Row-wise normalisation kernels around the HSTU projections (gfx950, HBM-bound).
//
// One wavefront owns one row at a time (grid-stride over rows): each lane loads its
// 16-byte pieces of the row once, keeps them in registers, reduces with wave shuffles,
// and writes once.  Weight gradients are accumulated per lane across all the rows a
// wave visits (lanes own fixed columns), reduced across the workgroup's waves through
// LDS and written as one fp32 partial per workgroup; a second small kernel sums the
// partials.  All math is fp32, outputs are cast to the I/O dtype, like the reference:
//   ops/pytorch/pt_layer_norm.py:24-38, ops/pytorch/pt_hstu_linear.py:23-65;
// kernels replaced: ops/triton/triton_layer_norm.py:77-309,
// ops/triton/triton_hstu_linear.py:48-337 (LN * u), :570-1036 (GroupNorm * u).
#include "hstu_common.cuh"
#include "capi_internal.h"

#ifndef NORM_NT
#define NORM_NT 2      // non-temporal hint on the row kernels' 16-byte loads (1) / stores (2) of 16-bit rows: stores on (layer norm fwd 81 -> 68 us,
                       // SiLU fwd 81 -> 62 us, the others -1..-3 %, layer step unchanged); loads mixed (norm_mul bwd +6 %): off.  profiles/r04_norm_nt.txt
#endif

#ifndef NORM_DROP_ROUNDS
#define NORM_DROP_ROUNDS 1     // multiply-xorshift rounds of the fused dropout's generator (norm_kernels.inc, drop_hash)
#endif
#ifndef NORM_FAST_SIGMOID
#define NORM_FAST_SIGMOID 1
#endif

namespace hstu {
// narrow / wide instances of every kernel and launcher (norm_kernels.inc)
namespace nw1 {
#define NORM_WIDE 1
#include "norm_kernels.inc"
#undef NORM_WIDE
}  // namespace nw1
namespace nw4 {
#define NORM_WIDE 4
#include "norm_kernels.inc"
#undef NORM_WIDE
}  // namespace nw4
// rows of up to 1024 elements (512 without 16-byte alignment) take the narrow instance
static bool norm_wide(int dim, const void* a, const void* b, const void* c, int elem_bytes) {
  const int v = 16 / elem_bytes;
  const bool aligned = dim % v == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0);
  return dim > (aligned ? 1024 : 512);
}
using nw1::kMaxNormBlocks;
using nw1::silu_launch;
}  // namespace hstu

using namespace hstu;

#define DISPATCH_DTYPE(dtype, CALL_BF16, CALL_F16, CALL_F32)                          \
  switch (dtype) {                                                                    \
    case HSTU_DTYPE_BF16: return CALL_BF16;                                           \
    case HSTU_DTYPE_F16: return CALL_F16;                                             \
    case HSTU_DTYPE_F32: return CALL_F32;                                             \
    default: return set_error(HSTU_EINVAL, "dtype must be bf16, fp16 or fp32");       \
  }

// (wide ? nw4::CALL : nw1::CALL): the row kernels' narrow or wide instance
#define NW(wide, ...) ((wide) ? nw4::__VA_ARGS__ : nw1::__VA_ARGS__)

extern "C" {

size_t hstu_norm_bwd_workspace_bytes(int64_t rows, int32_t dim) {
  (void)rows;
  return (size_t)kMaxNormBlocks * 2 * (size_t)dim * sizeof(float);
}

int hstu_layer_norm_fwd(const void* x, const void* weight, const void* bias, void* y, float* mean, float* rstd,
                        int64_t rows, int32_t dim, float eps, int dtype, void* stream) {
  if (rows == 0) return HSTU_OK;
  if (!x || !weight || !bias || !y) return set_error(HSTU_EINVAL, "layer_norm_fwd: NULL tensor");
  hipStream_t st = (hipStream_t)stream;
  const bool wd = norm_wide(dim, x, y, weight, dtype == HSTU_DTYPE_F32 ? 4 : 2);
  DISPATCH_DTYPE(dtype, NW(wd, ln_fwd<bf16_t>(x, weight, bias, y, mean, rstd, rows, dim, eps, st)),
                 NW(wd, ln_fwd<f16_t>(x, weight, bias, y, mean, rstd, rows, dim, eps, st)),
                 NW(wd, ln_fwd<float>(x, weight, bias, y, mean, rstd, rows, dim, eps, st)));
}

int hstu_layer_norm_bwd(const void* dy, const void* x, const void* weight, const float* mean, const float* rstd,
                        void* dx, float* dweight, float* dbias, float* partial_ws, int64_t rows, int32_t dim, int dtype,
                        void* stream) {
  return hstu_layer_norm_bwd_residual(dy, x, weight, mean, rstd, nullptr, dx, dweight, dbias, partial_ws, rows, dim, dtype, stream);
}

int hstu_layer_norm_bwd_residual(const void* dy, const void* x, const void* weight, const float* mean, const float* rstd,
                                 const void* dresidual, void* dx, float* dweight, float* dbias, float* partial_ws,
                                 int64_t rows, int32_t dim, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const void* dres = dresidual;
  if (!dweight || !dbias) return set_error(HSTU_EINVAL, "layer_norm_bwd: dweight/dbias are required");
  if (rows == 0) {
    (void)hipMemsetAsync(dweight, 0, dim * sizeof(float), st);
    (void)hipMemsetAsync(dbias, 0, dim * sizeof(float), st);
    return HSTU_OK;
  }
  if (!dy || !x || !weight || !mean || !rstd || !dx || !partial_ws) return set_error(HSTU_EINVAL, "layer_norm_bwd: NULL tensor");
  const bool wd = norm_wide(dim, dy, x, dx, dtype == HSTU_DTYPE_F32 ? 4 : 2);
  DISPATCH_DTYPE(dtype, NW(wd, ln_bwd<bf16_t>(dy, x, weight, mean, rstd, dx, dweight, dbias, partial_ws, rows, dim, dres, st)),
                 NW(wd, ln_bwd<f16_t>(dy, x, weight, mean, rstd, dx, dweight, dbias, partial_ws, rows, dim, dres, st)),
                 NW(wd, ln_bwd<float>(dy, x, weight, mean, rstd, dx, dweight, dbias, partial_ws, rows, dim, dres, st)));
}

// y = x * sigmoid(LayerNorm(x)): the gate in front of the preprocessors' and DlrmHSTU's MLPs (SwishLayerNorm)
int hstu_swish_layer_norm_fwd(const void* x, const void* weight, const void* bias, void* y, float* mean, float* rstd,
                              int64_t rows, int32_t dim, float eps, int dtype, void* stream) {
  if (rows == 0) return HSTU_OK;
  if (!x || !weight || !bias || !y) return set_error(HSTU_EINVAL, "swish_layer_norm_fwd: NULL tensor");
  hipStream_t st = (hipStream_t)stream;
  const bool wd = norm_wide(dim, x, y, weight, dtype == HSTU_DTYPE_F32 ? 4 : 2);
  DISPATCH_DTYPE(dtype, NW(wd, ln_fwd<bf16_t, true>(x, weight, bias, y, mean, rstd, rows, dim, eps, st)),
                 NW(wd, ln_fwd<f16_t, true>(x, weight, bias, y, mean, rstd, rows, dim, eps, st)),
                 NW(wd, ln_fwd<float, true>(x, weight, bias, y, mean, rstd, rows, dim, eps, st)));
}

int hstu_swish_layer_norm_bwd(const void* dy, const void* x, const void* weight, const void* bias, const float* mean,
                              const float* rstd, void* dx, float* dweight, float* dbias, float* partial_ws, int64_t rows,
                              int32_t dim, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!dweight || !dbias) return set_error(HSTU_EINVAL, "swish_layer_norm_bwd: dweight/dbias are required");
  if (rows == 0) {
    (void)hipMemsetAsync(dweight, 0, dim * sizeof(float), st);
    (void)hipMemsetAsync(dbias, 0, dim * sizeof(float), st);
    return HSTU_OK;
  }
  if (!dy || !x || !weight || !bias || !mean || !rstd || !dx || !partial_ws) return set_error(HSTU_EINVAL, "swish_layer_norm_bwd: NULL tensor");
  const bool wd = norm_wide(dim, dy, x, dx, dtype == HSTU_DTYPE_F32 ? 4 : 2);
  DISPATCH_DTYPE(dtype, NW(wd, ln_bwd<bf16_t, true>(dy, x, weight, mean, rstd, dx, dweight, dbias, partial_ws, rows, dim, nullptr, st, bias)),
                 NW(wd, ln_bwd<f16_t, true>(dy, x, weight, mean, rstd, dx, dweight, dbias, partial_ws, rows, dim, nullptr, st, bias)),
                 NW(wd, ln_bwd<float, true>(dy, x, weight, mean, rstd, dx, dweight, dbias, partial_ws, rows, dim, nullptr, st, bias)));
}

static int drop_ratio_ok(float r, const char* who) {
  if (!(r >= 0.f && r < 1.f)) return set_error(HSTU_EINVAL, "%s: dropout_ratio must be in [0, 1) (got %g)", who, (double)r);
  return HSTU_OK;
}

int hstu_norm_mul_dropout_fwd(const void* attn, const void* u, const void* weight, const void* bias, void* y, float* mean,
                              float* rstd, int64_t rows, int32_t heads, int32_t head_dim, float eps, int group_norm,
                              int concat_ux, float dropout_ratio, uint64_t seed, int dtype, void* stream) {
  return hstu_norm_mul_silu_fwd(attn, u, (int64_t)heads * head_dim, 0, weight, bias, y, mean, rstd, rows, heads, head_dim, eps,
                                group_norm, concat_ux, dropout_ratio, seed, dtype, stream);
}

int hstu_norm_mul_silu_fwd(const void* attn, const void* u, int64_t u_row_stride, int u_is_preactivation, const void* weight,
                           const void* bias, void* y, float* mean, float* rstd, int64_t rows, int32_t heads,
                           int32_t head_dim, float eps, int group_norm, int concat_ux, float dropout_ratio, uint64_t seed,
                           int dtype, void* stream) {
  if (int e = drop_ratio_ok(dropout_ratio, "norm_mul_dropout_fwd")) return e;
  if (u_row_stride < (int64_t)heads * head_dim) return set_error(HSTU_EINVAL, "norm_mul_fwd: u_row_stride is smaller than a row");
  const bool silu = u_is_preactivation != 0;
  if (rows == 0) return HSTU_OK;
  if (!attn || !u || !weight || !bias || !y) return set_error(HSTU_EINVAL, "norm_mul_fwd: NULL tensor");
  hipStream_t st = (hipStream_t)stream;
  const bool wd = norm_wide(heads * head_dim, attn, u, y, dtype == HSTU_DTYPE_F32 ? 4 : 2);
#define NM_FWD(T) (wd ? nw4::nm_fwd<T>(attn, u, weight, bias, y, mean, rstd, rows, heads, head_dim, eps, group_norm, concat_ux, nw4::make_drop_ctx(dropout_ratio, seed), u_row_stride, silu, st) \
                      : nw1::nm_fwd<T>(attn, u, weight, bias, y, mean, rstd, rows, heads, head_dim, eps, group_norm, concat_ux, nw1::make_drop_ctx(dropout_ratio, seed), u_row_stride, silu, st))
  DISPATCH_DTYPE(dtype, NM_FWD(bf16_t), NM_FWD(f16_t), NM_FWD(float));
#undef NM_FWD
}

int hstu_norm_mul_fwd(const void* attn, const void* u, const void* weight, const void* bias, void* y, float* mean,
                      float* rstd, int64_t rows, int32_t heads, int32_t head_dim, float eps, int group_norm,
                      int concat_ux, int dtype, void* stream) {
  return hstu_norm_mul_dropout_fwd(attn, u, weight, bias, y, mean, rstd, rows, heads, head_dim, eps, group_norm, concat_ux,
                                   0.f, 0, dtype, stream);
}

int hstu_norm_mul_bwd(const void* dy, const void* attn, const void* u, const void* weight, const void* bias,
                      const float* mean, const float* rstd, void* dattn, void* du, float* dweight, float* dbias,
                      float* partial_ws, int64_t rows, int32_t heads, int32_t head_dim, int group_norm, int concat_ux,
                      int dtype, void* stream) {
  return hstu_norm_mul_dropout_bwd(dy, attn, u, weight, bias, mean, rstd, dattn, du, dweight, dbias, partial_ws, rows, heads,
                                   head_dim, group_norm, concat_ux, 0.f, 0, dtype, stream);
}

int hstu_norm_mul_dropout_bwd(const void* dy, const void* attn, const void* u, const void* weight, const void* bias,
                              const float* mean, const float* rstd, void* dattn, void* du, float* dweight, float* dbias,
                              float* partial_ws, int64_t rows, int32_t heads, int32_t head_dim, int group_norm,
                              int concat_ux, float dropout_ratio, uint64_t seed, int dtype, void* stream) {
  const int64_t dim = (int64_t)heads * head_dim;
  return hstu_norm_mul_silu_bwd(dy, attn, u, dim, 0, weight, bias, mean, rstd, dattn, du, dim, dweight, dbias, partial_ws, rows,
                                heads, head_dim, group_norm, concat_ux, dropout_ratio, seed, dtype, stream);
}

int hstu_norm_mul_silu_bwd(const void* dy, const void* attn, const void* u, int64_t u_row_stride, int u_is_preactivation,
                           const void* weight, const void* bias, const float* mean, const float* rstd, void* dattn, void* du,
                           int64_t du_row_stride, float* dweight, float* dbias, float* partial_ws, int64_t rows,
                           int32_t heads, int32_t head_dim, int group_norm, int concat_ux, float dropout_ratio,
                           uint64_t seed, int dtype, void* stream) {
  if (int e = drop_ratio_ok(dropout_ratio, "norm_mul_dropout_bwd")) return e;
  if (u_row_stride < (int64_t)heads * head_dim || du_row_stride < (int64_t)heads * head_dim)
    return set_error(HSTU_EINVAL, "norm_mul_bwd: a row stride is smaller than a row");
  const bool silu = u_is_preactivation != 0;
  hipStream_t st = (hipStream_t)stream;
  const int width = group_norm ? heads : heads * head_dim;
  if (!dweight || !dbias) return set_error(HSTU_EINVAL, "norm_mul_bwd: dweight/dbias are required");
  if (rows == 0) {
    (void)hipMemsetAsync(dweight, 0, width * sizeof(float), st);
    (void)hipMemsetAsync(dbias, 0, width * sizeof(float), st);
    return HSTU_OK;
  }
  if (!dy || !attn || !u || !weight || !bias || !mean || !rstd || !dattn || !du || !partial_ws)
    return set_error(HSTU_EINVAL, "norm_mul_bwd: NULL tensor");
  const bool wd = norm_wide(heads * head_dim, attn, u, dy, dtype == HSTU_DTYPE_F32 ? 4 : 2) ||
                  norm_wide(heads * head_dim, dattn, du, nullptr, dtype == HSTU_DTYPE_F32 ? 4 : 2);
#define NM_BWD(T) (wd ? nw4::nm_bwd<T>(dy, attn, u, weight, bias, mean, rstd, dattn, du, dweight, dbias, partial_ws, rows, heads, head_dim, group_norm, concat_ux, nw4::make_drop_ctx(dropout_ratio, seed), u_row_stride, du_row_stride, silu, st) \
                      : nw1::nm_bwd<T>(dy, attn, u, weight, bias, mean, rstd, dattn, du, dweight, dbias, partial_ws, rows, heads, head_dim, group_norm, concat_ux, nw1::make_drop_ctx(dropout_ratio, seed), u_row_stride, du_row_stride, silu, st))
  DISPATCH_DTYPE(dtype, NM_BWD(bf16_t), NM_BWD(f16_t), NM_BWD(float));
#undef NM_BWD
}

int hstu_silu_fwd(const void* in, void* out, int64_t rows, int32_t cols, int64_t in_row_stride, int64_t out_row_stride,
                  int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DTYPE(dtype, (silu_launch<bf16_t, false>(nullptr, in, out, rows, cols, 0, in_row_stride, out_row_stride, st)),
                 (silu_launch<f16_t, false>(nullptr, in, out, rows, cols, 0, in_row_stride, out_row_stride, st)),
                 (silu_launch<float, false>(nullptr, in, out, rows, cols, 0, in_row_stride, out_row_stride, st)));
}

int hstu_silu_bwd(const void* dout, const void* in, void* din, int64_t rows, int32_t cols, int64_t dout_row_stride,
                  int64_t in_row_stride, int64_t din_row_stride, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DTYPE(dtype, (silu_launch<bf16_t, true>(dout, in, din, rows, cols, dout_row_stride, in_row_stride, din_row_stride, st)),
                 (silu_launch<f16_t, true>(dout, in, din, rows, cols, dout_row_stride, in_row_stride, din_row_stride, st)),
                 (silu_launch<float, true>(dout, in, din, rows, cols, dout_row_stride, in_row_stride, din_row_stride, st)));
}


int hstu_l2_norm_fwd(const void* x, void* y, int64_t rows, int32_t dim, float eps, int dtype, void* stream) {
  if (rows > 0 && (!x || !y)) return set_error(HSTU_EINVAL, "hstu_l2_norm_fwd: x and y must be non-NULL");
  hipStream_t st = (hipStream_t)stream;
  const bool wd = norm_wide(dim, x, y, nullptr, dtype == HSTU_DTYPE_F32 ? 4 : 2);
  DISPATCH_DTYPE(dtype, NW(wd, l2_launch<bf16_t, false>(x, nullptr, y, rows, dim, eps, st)),
                 NW(wd, l2_launch<f16_t, false>(x, nullptr, y, rows, dim, eps, st)),
                 NW(wd, l2_launch<float, false>(x, nullptr, y, rows, dim, eps, st)));
}

int hstu_l2_norm_bwd(const void* dy, const void* x, void* dx, int64_t rows, int32_t dim, float eps, int dtype, void* stream) {
  if (rows > 0 && (!x || !dy || !dx)) return set_error(HSTU_EINVAL, "hstu_l2_norm_bwd: dy, x and dx must be non-NULL");
  hipStream_t st = (hipStream_t)stream;
  const bool wd = norm_wide(dim, x, dy, dx, dtype == HSTU_DTYPE_F32 ? 4 : 2);
  DISPATCH_DTYPE(dtype, NW(wd, l2_launch<bf16_t, true>(x, dy, dx, rows, dim, eps, st)),
                 NW(wd, l2_launch<f16_t, true>(x, dy, dx, rows, dim, eps, st)),
                 NW(wd, l2_launch<float, true>(x, dy, dx, rows, dim, eps, st)));
}

}  // extern "C"
