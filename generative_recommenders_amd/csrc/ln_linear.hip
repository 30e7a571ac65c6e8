// C entry points of the fused LayerNorm + projection kernel (hstu_ln_linear.cuh).
#include <stdlib.h>

#include "hstu_ln_linear.cuh"

using namespace hstu;

extern "C" {

int hstu_ln_linear_fwd_supported(int64_t rows, int32_t k, int32_t n, int dtype) {
  (void)rows;
  return (dtype == HSTU_DTYPE_BF16 || dtype == HSTU_DTYPE_F16) && k == kLnlK && n > 0 && n % 32 == 0 && n <= kLnlMaxN;
}

int hstu_ln_linear_fwd(const void* x, int64_t ldx, const void* ln_weight, const void* ln_bias, float eps,
                       const void* w_nk, const void* bias, void* y, int64_t ldy, void* normed, int64_t ldn,
                       float* mean, float* rstd, int64_t rows, int32_t k, int32_t n, int dtype, void* stream) {
  if (rows == 0) return HSTU_OK;
  if (!x || !ln_weight || !ln_bias || !w_nk || !y) return set_error(HSTU_EINVAL, "ln_linear_fwd: NULL tensor");
  if (!hstu_ln_linear_fwd_supported(rows, k, n, dtype))
    return set_error(HSTU_EINVAL, "ln_linear_fwd: needs bf16 / fp16, k == %d, n a multiple of 32 up to %d (got k %d, n %d, dtype %d)",
                     kLnlK, kLnlMaxN, k, n, dtype);
  if (rows < 0 || ldx < k || ldy < n || (normed && ldn < k)) return set_error(HSTU_EINVAL, "ln_linear_fwd: bad rows / leading dimension");
  if ((ldx | ldy | (normed ? ldn : 0)) % 8 != 0 ||
      ((((uintptr_t)x | (uintptr_t)w_nk | (uintptr_t)y | (uintptr_t)normed)) & 15) != 0)
    return set_error(HSTU_EINVAL, "ln_linear_fwd: x, w, y and normed must be 16-byte aligned with leading dimensions that are multiples of 8");
  LnLinearArgs g;
  g.x = x; g.ln_w = ln_weight; g.ln_b = ln_bias; g.w = w_nk; g.bias = bias;
  g.y = y; g.normed = normed; g.mean = mean; g.rstd = rstd;
  g.rows = rows; g.ldx = ldx; g.ldy = ldy; g.ldn = ldn;
  g.n = n; g.n_tiles = n / 32;
  g.units = ((rows + kLnlBlockRows - 1) / kLnlBlockRows) * g.n_tiles;
  g.eps = eps;
  hipStream_t st = (hipStream_t)stream;
  return dtype == HSTU_DTYPE_BF16 ? launch_ln_linear<bf16_t>(g, st) : launch_ln_linear<f16_t>(g, st);
}

int hstu_linear_k512_supported(int64_t rows, int32_t k, int32_t n, int dtype) { return hstu_ln_linear_fwd_supported(rows, k, n, dtype); }

// y = x . w_nk^T (+ bias): the same kernel without the LayerNorm (rows of x held in registers, W streamed through LDS)
int hstu_linear_k512(const void* x, int64_t ldx, const void* w_nk, const void* bias, void* y, int64_t ldy,
                     int64_t rows, int32_t k, int32_t n, int dtype, void* stream) {
  if (rows == 0) return HSTU_OK;
  if (!x || !w_nk || !y) return set_error(HSTU_EINVAL, "linear_k512: NULL tensor");
  if (!hstu_linear_k512_supported(rows, k, n, dtype))
    return set_error(HSTU_EINVAL, "linear_k512: needs bf16 / fp16, k == %d, n a multiple of 32 up to %d (got k %d, n %d, dtype %d)",
                     kLnlK, kLnlMaxN, k, n, dtype);
  if (rows < 0 || ldx < k || ldy < n) return set_error(HSTU_EINVAL, "linear_k512: bad rows / leading dimension");
  if ((ldx | ldy) % 8 != 0 || ((((uintptr_t)x | (uintptr_t)w_nk | (uintptr_t)y)) & 15) != 0)
    return set_error(HSTU_EINVAL, "linear_k512: x, w and y must be 16-byte aligned with leading dimensions that are multiples of 8");
  LnLinearArgs g;
  g.x = x; g.ln_w = nullptr; g.ln_b = nullptr; g.w = w_nk; g.bias = bias;
  g.y = y; g.normed = nullptr; g.mean = nullptr; g.rstd = nullptr;
  g.rows = rows; g.ldx = ldx; g.ldy = ldy; g.ldn = 0;
  g.n = n; g.n_tiles = n / 32;
  g.units = ((rows + kLnlBlockRows - 1) / kLnlBlockRows) * g.n_tiles;
  g.eps = 0.f;
  hipStream_t st = (hipStream_t)stream;
  return dtype == HSTU_DTYPE_BF16 ? launch_ln_linear<bf16_t>(g, st) : launch_ln_linear<f16_t>(g, st);
}

}  // extern "C"
